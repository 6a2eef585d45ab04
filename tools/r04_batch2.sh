mkdir -p gpurun_out/r04
python tools/tree_path_probe.py 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_start_points.py tests/test_gpu_dispatch_table.py -q 2>&1 | tail -12
timeout 600 python -m pytest tests/test_gpu_catalog.py -q -k "ensembles_per_workgroup or band_pack or bit_identical" 2>&1 | tail -5
timeout 400 python tools/catalog_sizes.py --sizes 313,1250,2500,5000,10000,40000 --groups 0 --start kernel > gpurun_out/r04/catalog_sizes_stdform.jsonl 2>&1
ISOCHRONES_AMD_STD_PRIORS=0 timeout 400 python tools/catalog_sizes.py --sizes 313,1250,2500,5000,10000,40000 --groups 0 --start kernel > gpurun_out/r04/catalog_sizes_rtform.jsonl 2>&1
cut -c1-260 gpurun_out/r04/catalog_sizes_stdform.jsonl gpurun_out/r04/catalog_sizes_rtform.jsonl
for lib in default fw2 fw3; do
  if [ $lib = default ]; then unset ISOCHRONES_AMD_LIB; else export ISOCHRONES_AMD_LIB=$PWD/variants/libs/libiso_hip_$lib.so; fi
  timeout 900 python tools/sweep_fast_waves.py --reps 20 > gpurun_out/r04/fast_waves_$lib.jsonl 2>&1
done
unset ISOCHRONES_AMD_LIB
for wl in posterior prior_valid; do
  for lib in default noexp default noexp; do
    if [ $lib = default ]; then unset ISOCHRONES_AMD_LIB; else export ISOCHRONES_AMD_LIB=$PWD/variants/libs/libiso_hip_$lib.so; fi
    echo "$lib $wl $(timeout 300 python bench.py --no-extras --no-cpu-baseline --workload $wl --steps 100 2>/dev/null | python -c 'import json,sys; r=json.loads(sys.stdin.read()); print(r["ms_per_step"]*1e3)')" >> gpurun_out/r04/ab_fast_exp.txt
  done
done
unset ISOCHRONES_AMD_LIB
cat gpurun_out/r04/ab_fast_exp.txt
