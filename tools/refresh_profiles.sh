# Regenerate the files under profiles/<round>/ on the GPU box:  bash tools/refresh_profiles.sh r02
set -x
R=${1:-r02}
OUT=gpurun_out/$R
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $OUT
python bench.py > $OUT/bench_cfg2_1gpu.json 2> $OUT/bench.err
python bench_configs.py --configs cfg1,cfg3,cfg4,cfg5,tree,primitives,astero,nested,published > $OUT/bench_configs_1gpu.jsonl 2> $OUT/bench_configs.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/prof -- python $ROOT/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras > $ROOT/$OUT/prof_bench.json 2> $ROOT/$OUT/prof.err
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/prof_all -- python $ROOT/bench_configs.py --configs cfg3,cfg4,cfg5,tree,primitives,astero > $ROOT/$OUT/prof_all.jsonl 2> $ROOT/$OUT/prof_all.err
cd $ROOT
for d in prof prof_all; do
  f=$(find $OUT/$d -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" $OUT/kernel_stats_$d.csv
  find $OUT/$d -name "*.csv" -size +1M -delete
done
tail -c 1500 $OUT/bench_cfg2_1gpu.json
