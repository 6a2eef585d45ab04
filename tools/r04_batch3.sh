mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_gpu_start_points.py tests/test_gpu_dispatch_table.py -q 2>&1 | tail -8
timeout 2400 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_dispatch_table.py --deselect tests/test_gpu_start_points.py 2>&1 | tail -8
(timeout 300 python tests/soak/soak_sampler.py 150 41 2>&1 | tail -3; timeout 200 python tests/soak/soak.py 90 42 2>&1 | tail -2; timeout 150 python tests/soak/soak_tree.py 60 43 2>&1 | tail -2) > gpurun_out/r04/soak_mid.txt 2>&1
cat gpurun_out/r04/soak_mid.txt | cut -c1-300
