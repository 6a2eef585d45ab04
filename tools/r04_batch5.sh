mkdir -p gpurun_out/r04
for lib in default q3 default q3; do
  if [ $lib = default ]; then unset ISOCHRONES_AMD_LIB; else export ISOCHRONES_AMD_LIB=$PWD/variants/libs/libiso_hip_$lib.so; fi
  echo "$lib $(timeout 300 python tools/quantile_timing.py 2>/dev/null | tail -1)" >> gpurun_out/r04/quantile_waves_ab.txt
done
unset ISOCHRONES_AMD_LIB
python - <<'PY'
import json
for ln in open("gpurun_out/r04/quantile_waves_ab.txt"):
    lib, js = ln.split(" ", 1)
    r = json.loads(js)
    print(lib, {k: round(v["auto"]["ms"], 3) for k, v in r.items()})
PY
timeout 900 python -m pytest tests/test_gpu_bench_launch.py tests/test_gpu_parity.py -q -k "bench or library_built" 2>&1 | tail -4
