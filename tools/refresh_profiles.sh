set -x
mkdir -p gpurun_out/final
python bench.py > gpurun_out/final/bench_cfg2_1gpu.json 2> gpurun_out/final/bench.err
python bench_configs.py --configs cfg1,cfg3,cfg4,cfg5,tree,primitives,astero,nested,published > gpurun_out/final/bench_configs_1gpu.jsonl 2> gpurun_out/final/bench_configs.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/final/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras > $GRAFT_REPO_ROOT/gpurun_out/final/prof_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/final/prof.err
cd $GRAFT_REPO_ROOT
find gpurun_out/final/prof -name "*kernel_stats.csv" | head
tail -c 600 gpurun_out/final/bench_cfg2_1gpu.json
