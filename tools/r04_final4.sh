# after the raw min / max of the quantile kernel and the grouped-form test: the whole GPU suite, quantile soak + timing
OUT=gpurun_out/r04j; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -6 > $OUT/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1
timeout 200 python tests/soak/soak_quantiles.py 120 131 2>&1 | tail -1 > $OUT/soak_quantiles.txt
for i in 1 2; do
  echo -n "new " >> $OUT/quantile_rawminmax_ab.txt; timeout 200 python tools/quantile_timing.py >> $OUT/quantile_rawminmax_ab.txt 2>> $OUT/err.txt
  echo -n "old " >> $OUT/quantile_rawminmax_ab.txt; ISOCHRONES_AMD_LIB=variants/libs/libiso_hip_qold.so timeout 200 python tools/quantile_timing.py >> $OUT/quantile_rawminmax_ab.txt 2>> $OUT/err.txt
done
cat $OUT/pytest_gpu.txt $OUT/smoke.txt $OUT/soak_quantiles.txt
