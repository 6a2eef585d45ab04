# Round-4 profile set on the GPU box:  bash tools/r04_final_profiles.sh   (results under gpurun_out/r04f, copied to profiles/r04 by hand)
R=r04f
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
# per-launch means of the headline kernel in the profiled run: pre-roll / warm-up / timed / rest
python - <<'PY' > gpurun_out/r04f/kernel_trace_timed_launches.txt 2>&1
import csv, glob
f = glob.glob("gpurun_out/r04f/prof/**/*kernel_trace.csv", recursive=True)
rows = []
for path in f:
    for r in csv.DictReader(open(path)):
        if "k_lnpost_fast<0, 1, 1, false, false>" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
rows.sort()
d = [x[1] / 1e3 for x in rows]
big = [x for x in d if x > 40.0]           # the 10^6-row launches (the first small one builds the packs)
def mean(v): return sum(v) / max(len(v), 1)
print("launches of k_lnpost_fast<0, 1, 1, false, false> with 10^6 rows: %d" % len(big))
print("pre-roll (400): %.2f us   warm-up (20): %.2f us   timed rotating (200): %.2f us   one batch repeated (rest, %d): %.2f us"
      % (mean(big[:400]), mean(big[400:420]), mean(big[420:620]), len(big) - 620, mean(big[620:])))
PY
find $OUT -name "*.csv" -size +1M -delete
bash tools/pmc_collect.sh $OUT/pmc > $OUT/pmc_collect.log 2>&1
cp $OUT/pmc/pmc_summary.json $OUT/pmc_summary.json 2>/dev/null; cp $OUT/pmc/manifest.json $OUT/pmc_manifest.json 2>/dev/null
rm -rf $OUT/pmc/pass*/ 2>/dev/null
python tools/odd_shape_timing.py > $OUT/odd_shapes_timing.json 2> $OUT/odd.err
python tools/sampler_mode_sweep.py > $OUT/sampler_mode_sweep.txt 2>&1
( timeout 500 python tests/soak/soak.py 300 101 2>&1 | tail -1; timeout 400 python tests/soak/soak_tree.py 200 102 2>&1 | tail -1; timeout 200 python tests/soak/soak_quantiles.py 60 103 2>&1 | tail -1; timeout 700 python tests/soak/soak_sampler.py 480 104 2>&1 | tail -1; SOAK_KIND=iso SOAK_NSTARS=3 SOAK_NB=9 SOAK_CATALOG_FRACTION=0 timeout 200 python tests/soak/soak_sampler.py 100 105 2>&1 | tail -1; SOAK_KIND=iso SOAK_NSTARS=3 SOAK_NB=4 SOAK_CATALOG_FRACTION=0 timeout 200 python tests/soak/soak_sampler.py 60 106 2>&1 | tail -1; SOAK_CATALOG_FRACTION=1 timeout 300 python tests/soak/soak_sampler.py 180 107 2>&1 | tail -1; timeout 200 python tests/soak/soak_primitives.py 60 108 2>&1 | tail -1 ) > $OUT/soak.txt 2>&1
tail -c 600 $OUT/bench_cfg2_1gpu_driver_args.json; cat $OUT/kernel_trace_timed_launches.txt; cat $OUT/soak.txt | cut -c1-260
