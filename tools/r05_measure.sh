# Round-5 measurements on the GPU box (one script, parametrised by stage): bash tools/r05_measure.sh <stage> [...]
#   tests      the dispatch-table closure + new sampler tests
#   fits       bench_configs fits,cfg4,tree (any-model sampler wall-clocks), mailbox latency
#   bench      bench.py default line + the driver's arguments
#   shapes     one star's fit by model shape, catalog fit by size, ns per move by walkers
#   prof       rocprofv3 --kernel-trace --stats of bench.py and of the fits
# Results under gpurun_out/r05/ (copy what should be judged into profiles/r05/).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05
mkdir -p $OUT
for stage in "$@"; do
case $stage in
tests)
  timeout 1500 python -m pytest tests/test_gpu_dispatch_table.py tests/test_gpu_sampler_any.py -q 2>&1 | tail -15 | tee $OUT/pytest_dispatch_and_any.txt ;;
suite)
  timeout 2500 python -m pytest tests -m gpu -q 2>&1 | tail -8 | tee $OUT/pytest_gpu_suite.txt ;;
fits)
  python bench_configs.py --configs fits,cfg4,tree > $OUT/bench_configs_fits.jsonl 2> $OUT/bench_configs_fits.err; tail -c 2500 $OUT/bench_configs_fits.jsonl
  python tools/mailbox_latency.py 2>&1 | grep -v amdgpu.ids | tee $OUT/mailbox_latency.txt
  python tools/scalar_latency.py 2>&1 | grep -v amdgpu.ids | head -9 | tee $OUT/scalar_latency.txt ;;
bench)
  python bench.py > $OUT/bench_cfg2_1gpu.json 2> $OUT/bench.err; tail -c 600 $OUT/bench_cfg2_1gpu.json
  python bench.py --steps 20 --warmup 5 > $OUT/bench_cfg2_1gpu_driver_args.json 2>> $OUT/bench.err ;;
shapes)
  python tools/single_fit_shapes.py 2>/dev/null | grep "^{" > $OUT/single_fit_shapes.jsonl; tail -3 $OUT/single_fit_shapes.jsonl | cut -c1-300
  python tools/catalog_sizes.py --sizes 313,625,1250,2500,5000,10000 2>/dev/null | grep "^{" > $OUT/catalog_sizes.jsonl; tail -2 $OUT/catalog_sizes.jsonl | cut -c1-300
  python tools/walker_packing_probe.py 2>/dev/null | grep "^{" > $OUT/walker_packing.jsonl; cat $OUT/walker_packing.jsonl | cut -c1-200 ;;
prof)
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python $ROOT/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras > $OUT/bench_profiled_run.json 2> $OUT/prof.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_fits -- python $ROOT/bench_configs.py --configs fits,cfg4 > $OUT/prof_fits.jsonl 2> $OUT/prof_fits.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_catalog -- python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/prof_catalog.json 2> $OUT/prof_catalog.err
  cd $ROOT
  for d in prof prof_fits prof_catalog; do
    f=$(find $OUT/$d -name "*kernel_stats.csv" | head -1)
    [ -n "$f" ] && cp "$f" $OUT/kernel_stats_$d.csv
    find $OUT/$d -name "*.csv" -size +1M -delete
  done
  head -5 $OUT/kernel_stats_prof.csv; head -12 $OUT/kernel_stats_prof_fits.csv; head -16 $OUT/kernel_stats_prof_catalog.csv ;;
esac
done
