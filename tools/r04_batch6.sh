#!/bin/bash
# round 4, last session: quantile kernel (lean binning, scalar-base loads) against the previous one, wide-pack interp forms
O=gpurun_out/r04b; mkdir -p $O
for i in 1 2; do
  echo -n "new " >> $O/quantile_lean_ab.txt; timeout 200 python tools/quantile_timing.py >> $O/quantile_lean_ab.txt 2>> $O/err.txt
  echo -n "old " >> $O/quantile_lean_ab.txt; ISOCHRONES_AMD_LIB=variants/libs/libiso_hip_qold.so timeout 200 python tools/quantile_timing.py >> $O/quantile_lean_ab.txt 2>> $O/err.txt
done
timeout 300 python tools/wide_form_ab.py > $O/wide_form_ab.jsonl 2>> $O/err.txt
timeout 400 python -m pytest tests/test_gpu_dispatch_table.py tests/test_gpu_catalog.py -x -q 2>&1 | tail -6 > $O/pytest_subset.txt
cat $O/pytest_subset.txt; tail -5 $O/err.txt
