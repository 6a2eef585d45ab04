#!/bin/bash
# quantile kernel: ensemble-fastest wave order (default) against parameter-fastest (ISOCHRONES_AMD_QUANTILE_ORDER=0) and
# against the previous library; wide-pack interp defaults
O=gpurun_out/r04b; mkdir -p $O
for i in 1 2; do
  echo -n "ens_fastest " >> $O/quantile_order_ab.txt; timeout 200 python tools/quantile_timing.py >> $O/quantile_order_ab.txt 2>> $O/err.txt

  echo -n "old_library " >> $O/quantile_order_ab.txt; ISOCHRONES_AMD_LIB=variants/libs/libiso_hip_qold.so timeout 200 python tools/quantile_timing.py >> $O/quantile_order_ab.txt 2>> $O/err.txt
done
timeout 300 python tools/wide_form_ab.py > $O/wide_form_ab2.jsonl 2>> $O/err.txt
timeout 400 python -m pytest tests/test_gpu_dispatch_table.py tests/test_gpu_catalog.py -x -q 2>&1 | tail -3 > $O/pytest_subset2.txt
cat $O/pytest_subset2.txt
