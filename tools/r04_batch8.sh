#!/bin/bash
# quantile kernel: lane groups of a register S steps apart (default) against consecutive steps (ISOCHRONES_AMD_QUANTILE_SPREAD=0),
# on independent values and on chains that repeat a walker's value with probability 0.7; exactness tests + soak
O=gpurun_out/r04l; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_dispatch_table.py tests/test_gpu_catalog.py tests/test_gpu_parity.py -q -x 2>&1 | tail -3 > $O/pytest_subset.txt
timeout 200 python tests/soak/soak_quantiles.py 100 141 2>&1 | tail -1 > $O/soak_quantiles.txt
for i in 1 2; do
  for mode in "" "--correlated"; do
    echo -n "spread$mode " >> $O/quantile_spread_ab.txt; timeout 200 python tools/quantile_timing.py $mode >> $O/quantile_spread_ab.txt 2>> $O/err.txt
    echo -n "consecutive$mode " >> $O/quantile_spread_ab.txt; ISOCHRONES_AMD_QUANTILE_SPREAD=0 timeout 200 python tools/quantile_timing.py $mode >> $O/quantile_spread_ab.txt 2>> $O/err.txt
  done
done
cat $O/pytest_subset.txt $O/soak_quantiles.txt
