# Round-4 closing set after the wide-pack interp forms and the lean quantile kernel (batch / sampler kernels unchanged since
# tools/r04_final2.sh: their sweeps and PMC passes stand):   bash tools/r04_final3.sh   (results under gpurun_out/r04i)
R=r04i
OUT=gpurun_out/$R
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > $OUT/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1
( timeout 200 python tests/soak/soak_quantiles.py 120 121 2>&1 | tail -1; timeout 200 python tests/soak/soak_primitives.py 120 122 2>&1 | tail -1; timeout 150 python tests/soak/soak.py 60 123 2>&1 | tail -1 ) > $OUT/soak_final3.txt 2>&1
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
python - <<'PY' > gpurun_out/r04i/kernel_trace_timed_launches.txt 2>&1
import csv, glob
f = glob.glob("gpurun_out/r04i/prof/**/*kernel_trace.csv", recursive=True)
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
cat $OUT/pytest_gpu.txt $OUT/smoke.txt; tail -c 300 $OUT/bench_cfg2_1gpu_driver_args.json; cat $OUT/kernel_trace_timed_launches.txt; cat $OUT/soak_final3.txt | cut -c1-260
