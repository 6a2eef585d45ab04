#!/bin/bash
# quantile kernel with four of a lane's 50 values in LDS (no spill) against the kernel one commit earlier (qhead) and the
# round's starting kernel (qold); exactness tests + soak first
O=gpurun_out/r04n; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_dispatch_table.py tests/test_gpu_catalog.py -q -x 2>&1 | tail -3 > $O/pytest_subset.txt
timeout 200 python tests/soak/soak_quantiles.py 100 151 2>&1 | tail -1 > $O/soak_quantiles.txt
for i in 1 2; do
  for mode in "" "--correlated"; do
    echo -n "stash$mode " >> $O/quantile_stash_ab.txt; timeout 200 python tools/quantile_timing.py $mode >> $O/quantile_stash_ab.txt 2>> $O/err.txt
    echo -n "head$mode " >> $O/quantile_stash_ab.txt; ISOCHRONES_AMD_LIB=variants/libs/libiso_hip_qhead.so timeout 200 python tools/quantile_timing.py $mode >> $O/quantile_stash_ab.txt 2>> $O/err.txt
  done
done
cat $O/pytest_subset.txt $O/soak_quantiles.txt
