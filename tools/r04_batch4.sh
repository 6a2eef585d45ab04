mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_gpu_dispatch_table.py -q 2>&1 | tail -4
timeout 3000 python -m pytest tests -m gpu -q --deselect tests/test_gpu_dispatch_table.py --deselect tests/test_gpu_start_points.py 2>&1 | tail -15
for p in auto generic; do echo "path $p: $(python bench.py --path $p --no-extras --no-cpu-baseline --steps 100 2>/dev/null | python -c 'import json,sys; r=json.loads(sys.stdin.read()); print(r["ms_per_step"]*1e3, "us")')"; done > gpurun_out/r04/generic_kernel_timing.txt 2>&1
for wl in prior posterior; do echo "generic $wl: $(python bench.py --path generic --workload $wl --no-extras --no-cpu-baseline --steps 100 2>/dev/null | python -c 'import json,sys; r=json.loads(sys.stdin.read()); print(r["ms_per_step"]*1e3, "us")')"; done >> gpurun_out/r04/generic_kernel_timing.txt 2>&1
echo "tree register form: $(python bench_configs.py --configs tree 2>/dev/null | tail -1)" > gpurun_out/r04/tree_kernel_timing.txt
echo "tree runtime-leaf form: $(ISOCHRONES_AMD_TREE_RUNTIME_LEAVES=1 python bench_configs.py --configs tree 2>/dev/null | tail -1)" >> gpurun_out/r04/tree_kernel_timing.txt
echo "tree generic kernel: $(ISOCHRONES_AMD_PATH=generic python bench_configs.py --configs tree 2>/dev/null | tail -1)" >> gpurun_out/r04/tree_kernel_timing.txt
cat gpurun_out/r04/generic_kernel_timing.txt gpurun_out/r04/tree_kernel_timing.txt | cut -c1-300
