# Round-4 profile set after the single-binary kernel (reduced: no PMC passes - the batch kernels' source did not change):
#   bash tools/r04_final2.sh     (results under gpurun_out/r04h, copied to profiles/r04 by hand)
R=r04h
OUT=gpurun_out/$R
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $OUT
python bench.py > $OUT/bench_cfg2_1gpu.json 2> $OUT/bench.err
python bench.py --steps 20 --warmup 5 > $OUT/bench_cfg2_1gpu_driver_args.json 2> $OUT/bench_driver.err
python bench_configs.py --configs cfg1,cfg3,cfg4,cfg5,tree,primitives,astero,nested,published > $OUT/bench_configs_1gpu.jsonl 2> $OUT/bench_configs.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/prof -- python $ROOT/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras > $ROOT/$OUT/bench_profiled_run.json 2> $ROOT/$OUT/prof.err
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/prof_all -- python $ROOT/bench_configs.py --configs cfg3,cfg4,cfg5,tree,primitives,astero > $ROOT/$OUT/prof_all.jsonl 2> $ROOT/$OUT/prof_all.err
cd $ROOT
for d in prof prof_all; do
  f=$(find $OUT/$d -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" $OUT/kernel_stats_$d.csv
done
python - <<'PY' > gpurun_out/r04h/kernel_trace_timed_launches.txt 2>&1
import csv, glob
f = glob.glob("gpurun_out/r04h/prof/**/*kernel_trace.csv", recursive=True)
rows = []
for path in f:
    for r in csv.DictReader(open(path)):
        if "k_lnpost_fast<0, 1, 1, false, false>" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
rows.sort()
d = [x[1] / 1e3 for x in rows]
big = [x for x in d if x > 40.0]
def mean(v): return sum(v) / max(len(v), 1)
print("launches of k_lnpost_fast<0, 1, 1, false, false> with 10^6 rows: %d" % len(big))
print("pre-roll (400): %.2f us   warm-up (20): %.2f us   timed rotating (200): %.2f us   one batch repeated (rest, %d): %.2f us"
      % (mean(big[:400]), mean(big[400:420]), mean(big[420:620]), len(big) - 620, mean(big[620:])))
PY
find $OUT -name "*.csv" -size +1M -delete
python tools/sampler_mode_sweep.py > $OUT/sampler_mode_sweep.txt 2>&1
( timeout 300 python tests/soak/soak.py 180 111 2>&1 | tail -1; timeout 200 python tests/soak/soak_tree.py 90 112 2>&1 | tail -1; timeout 500 python tests/soak/soak_sampler.py 360 114 2>&1 | tail -1; SOAK_KIND=iso SOAK_NSTARS=2 SOAK_NB=11 SOAK_CATALOG_FRACTION=0 timeout 200 python tests/soak/soak_sampler.py 60 115 2>&1 | tail -1; SOAK_KIND=iso SOAK_NSTARS=3 SOAK_NB=9 SOAK_CATALOG_FRACTION=0 timeout 200 python tests/soak/soak_sampler.py 60 116 2>&1 | tail -1; SOAK_CATALOG_FRACTION=1 timeout 200 python tests/soak/soak_sampler.py 90 117 2>&1 | tail -1 ) > $OUT/soak.txt 2>&1
tail -c 400 $OUT/bench_cfg2_1gpu_driver_args.json; cat $OUT/kernel_trace_timed_launches.txt; cat $OUT/soak.txt | cut -c1-260
