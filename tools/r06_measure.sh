# Round-6 measurements on the GPU box (one script, parametrised by stage): bash tools/r06_measure.sh <stage> [...]
#   triple     the one-star-per-row sampler of triples: dispatch-table tests of the iso3 family + time per step by shape
#   replay     the reference-shape catalog (300 walkers, isochrones; singles and binaries) replayed against the oracle
#   tree       tree evaluator: tests of the tree kernels + fits (resolved binary 256 x 5 000) + batch kernel timing
#   tests      dispatch-table closure + sampler tests
#   suite      the whole GPU suite
#   fits       bench_configs fits,cfg4,tree; mailbox / scalar latency
#   bench      bench.py default line + the driver's arguments
#   shapes     one star's fit by model shape, catalog fit by size
#   prof       rocprofv3 --kernel-trace --stats of bench.py and of the fits
# Results under gpurun_out/r06/ (copy what should be judged into profiles/r06/).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06
mkdir -p $OUT
for stage in "$@"; do
case $stage in
triple)
  timeout 1500 python -m pytest tests/test_gpu_dispatch_table.py -q -x -k "iso-3 or iso3" 2>&1 | tail -15 | tee $OUT/pytest_triple.txt
  SHAPES=iso:2:6,iso:3:3,iso:3:6,iso:3:9,iso:3:12 python tools/single_fit_shapes.py 2>/dev/null | grep "^{" > $OUT/single_fit_shapes_triple.jsonl; cut -c1-200 $OUT/single_fit_shapes_triple.jsonl
  SHAPES=iso:3:9 ISOCHRONES_AMD_STAR_LANES=0 python tools/single_fit_shapes.py 2>/dev/null | grep "^{" > $OUT/single_fit_shapes_triple_off.jsonl; cut -c1-200 $OUT/single_fit_shapes_triple_off.jsonl
  SHAPES=iso:3:9 python tools/single_fit_shapes.py 300 2000 2>/dev/null | grep "^{" >> $OUT/single_fit_shapes_triple.jsonl; tail -1 $OUT/single_fit_shapes_triple.jsonl | cut -c1-200
  SHAPES=iso:3:9 ISOCHRONES_AMD_STAR_LANES=0 python tools/single_fit_shapes.py 300 2000 2>/dev/null | grep "^{" >> $OUT/single_fit_shapes_triple_off.jsonl; tail -1 $OUT/single_fit_shapes_triple_off.jsonl | cut -c1-200 ;;
triple_sweep)
  # one star per row against one lane per triple, by walkers (moves per half-step = walkers / 2) and bands
  for W in 32 64 128 256 300; do
    for TM in 0 1000; do
      SHAPES=iso:3:3,iso:3:9 ISOCHRONES_AMD_TRIPLE_MOVES=$TM python tools/single_fit_shapes.py $W 2000 3 2>/dev/null | grep "^{" | sed "s/\"lib\": \"default\"/\"triple_moves\": $TM/" >> $OUT/triple_sweep.jsonl
    done
  done
  python - <<PY
import json
for l in open("$OUT/triple_sweep.jsonl"):
    d = json.loads(l)
    print(d["n_bands"], d["walkers"], d["triple_moves"], round(d["us_per_step"], 2), d["lnprob_crc"])
PY
  ;;
tree_ab)
  # variant libraries of the tree units (tools/build_variant.py --only iso_fast_tree,iso_fast_stretch_tree): batch kernel + fit
  for rep in 1 2; do
  for name in ${TREE_LIBS:-default bm0 bm1}; do
    if [ "$name" = default ]; then unset ISOCHRONES_AMD_LIB; else export ISOCHRONES_AMD_LIB=$ROOT/variants/libs/libiso_hip_$name.so; fi
    python bench_configs.py --configs fits,tree 2>/dev/null | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    if d['config'] == 'fits': print('$name', 'fit us/step', round(d['tree_resolved_binary']['us_per_step'], 2), 'acc', d['tree_resolved_binary']['acceptance'], '300x300 ms', round(1e3 * d['tree_resolved_binary']['fit_mcmc_300x300_wall_s'], 2))
    else: print('$name', 'batch us', round(1e3 * d['kernel_ms'], 1), 'rel', d['parity_max_rel_err'])
" | tee -a $OUT/tree_ab.txt
  done
  done
  unset ISOCHRONES_AMD_LIB ;;
catalog)
  python tools/catalog_sizes.py --sizes 313,1250,10000 2>/dev/null | grep "^{" | tee -a $OUT/catalog_sizes.jsonl | cut -c1-400
  python tools/ab_kernels.py --cases cfg4,cfg5,cfg5ref --label default 2>/dev/null | grep '^{' | tee -a $OUT/ab_default.jsonl | cut -c1-1500 ;;
catalog_tests)
  timeout 2400 python -m pytest tests/test_gpu_catalog.py tests/test_gpu_sampler_oracle.py tests/test_gpu_start_points.py -q -x 2>&1 | tail -8 | tee $OUT/pytest_catalog.txt ;;
service)
  timeout 1500 python -m pytest tests/test_gpu_dispatch_table.py -q -x -k "interpolation or eep_unit" 2>&1 | tail -15 | tee $OUT/pytest_service.txt
  python tools/scalar_latency.py 2>&1 | grep -v amdgpu.ids | head -11 | tee $OUT/scalar_latency.txt
  ISOCHRONES_AMD_MAILBOX=0 python tools/scalar_latency.py 2>&1 | grep -v amdgpu.ids | head -11 | tee $OUT/scalar_latency_launch_path.txt ;;
waves)
  timeout 1500 python -m pytest tests/test_gpu_resident_waves.py tests/test_gpu_sampler_any.py -q -x 2>&1 | tail -12 | tee $OUT/pytest_waves.txt
  timeout 1500 python -m pytest tests/test_gpu_dispatch_table.py -q -x -k "tree or interpolation or eep_unit" 2>&1 | tail -12 | tee -a $OUT/pytest_waves.txt ;;
groups)
  python tools/catalog_sizes.py --sizes 313,1250,2500 --groups 0,2,4,8 2>/dev/null | grep "^{" | tee -a $OUT/catalog_groups.jsonl | cut -c1-330 ;;
bcast)
  python tools/broadcast_ab.py 3 2>/dev/null | grep "^{" | tee $OUT/broadcast_ab.jsonl
  timeout 900 python -m pytest tests/test_gpu_rccl.py -q -x 2>&1 | tail -5 | tee $OUT/pytest_rccl.txt ;;
pmc)
  bash tools/pmc_collect.sh gpurun_out/r06/pmc ${PMC_CASES:-cfg2,cfg3,generic,tree,tree_generic} 2>&1 | tail -12
  python - <<PY
import json
d = json.load(open("$OUT/pmc/pmc_summary.json"))
for k, v in (d.items() if isinstance(d, dict) else enumerate(d)):
    print(k, json.dumps(v)[:400])
PY
  ;;
msl)
  for rep in 1 2; do
  for name in default msl14; do
    if [ "$name" = default ]; then unset ISOCHRONES_AMD_LIB; else export ISOCHRONES_AMD_LIB=$ROOT/variants/libs/libiso_hip_$name.so; fi
    python tools/catalog_sizes.py --sizes 313,1250,10000 2>/dev/null | grep "^{" | sed "s/^{/{\"lib\": \"$name\", /" | tee -a $OUT/catalog_msl.jsonl | cut -c1-300
    python tools/ab_kernels.py --cases cfg5,cfg5ref --label $name 2>/dev/null | grep '^{' | tee -a $OUT/ab_msl.jsonl | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    print(d['label'], {k: (round(v['wall_s_min'] * 1e3, 2), v['rows_digest']) for k, v in d.items() if isinstance(v, dict) and 'wall_s_min' in v})
"
  done
  done
  unset ISOCHRONES_AMD_LIB ;;
qbig)
  # k_chain_quantiles_big on the chains of a reference-shape catalog fit (QBIG_LIBS: variant libraries to put beside the default)
  timeout 900 python -m pytest tests/test_gpu_catalog.py -q -k "chain_quantiles" 2>&1 | tail -3
  for rep in 1 2; do
  for name in ${QBIG_LIBS:-default}; do
    if [ "$name" = default ]; then unset ISOCHRONES_AMD_LIB; else export ISOCHRONES_AMD_LIB=$ROOT/variants/libs/libiso_hip_$name.so; fi
    [ "$name" = default ] || [ -f "$ISOCHRONES_AMD_LIB" ] || continue
    python tools/quantile_big_timing.py --label $name 2>/dev/null | grep '^{' | tee -a $OUT/quantile_big_ab.jsonl
  done
  done
  unset ISOCHRONES_AMD_LIB ;;
dispatch)
  timeout 2400 python -m pytest tests/test_gpu_dispatch_table.py -q 2>&1 | tail -12 | tee $OUT/pytest_dispatch.txt ;;
soak)
  # randomised GPU-vs-oracle runs on the round's final library (tests/soak): seeds of this round
  python tests/soak/soak_sampler.py ${SOAK_S:-400} 61 2>/dev/null | tail -1 | tee $OUT/soak.txt
  python tests/soak/soak.py 200 62 2>/dev/null | tail -1 | tee -a $OUT/soak.txt
  python tests/soak/soak_tree.py 150 63 2>/dev/null | tail -1 | tee -a $OUT/soak.txt
  python tests/soak/soak_primitives.py 100 64 2>/dev/null | tail -1 | tee -a $OUT/soak.txt ;;
replay)
  timeout 1500 python -m pytest tests/test_gpu_sampler_oracle.py -q -x -k "reference_shape" 2>&1 | tail -15 | tee $OUT/pytest_replay.txt ;;
tree)
  timeout 1500 python -m pytest tests/test_gpu_sampler_any.py tests/test_gpu_dispatch_table.py -q -x -k "tree or Tree" 2>&1 | tail -15 | tee $OUT/pytest_tree.txt
  python bench_configs.py --configs fits,tree > $OUT/bench_configs_fits.jsonl 2> $OUT/bench_configs_fits.err; tail -c 2500 $OUT/bench_configs_fits.jsonl ;;
tests)
  timeout 2400 python -m pytest tests/test_gpu_dispatch_table.py tests/test_gpu_sampler_any.py -q 2>&1 | tail -15 | tee $OUT/pytest_dispatch_and_any.txt ;;
suite)
  timeout 3000 python -m pytest tests -m gpu -q > $OUT/pytest_gpu_suite_full.txt 2>&1; tail -8 $OUT/pytest_gpu_suite_full.txt | tee $OUT/pytest_gpu_suite.txt
  grep -q " failed" $OUT/pytest_gpu_suite.txt || rm -f $OUT/pytest_gpu_suite_full.txt ;;      # (the whole log only when something failed)
fits)
  python bench_configs.py --configs fits,cfg4,tree > $OUT/bench_configs_fits.jsonl 2> $OUT/bench_configs_fits.err; tail -c 2500 $OUT/bench_configs_fits.jsonl
  python tools/mailbox_latency.py 2>&1 | grep -v amdgpu.ids | tee $OUT/mailbox_latency.txt
  python tools/scalar_latency.py 2>&1 | grep -v amdgpu.ids | head -9 | tee $OUT/scalar_latency.txt ;;
bench)
  python bench.py > $OUT/bench_cfg2_1gpu.json 2> $OUT/bench.err; head -c 1500 $OUT/bench_cfg2_1gpu.json
  python bench.py --steps 20 --warmup 5 > $OUT/bench_cfg2_1gpu_driver_args.json 2>> $OUT/bench.err ;;
shapes)
  python tools/single_fit_shapes.py 2>/dev/null | grep "^{" > $OUT/single_fit_shapes.jsonl; tail -3 $OUT/single_fit_shapes.jsonl | cut -c1-300
  python tools/catalog_sizes.py --sizes 313,625,1250,2500,5000,10000 2>/dev/null | grep "^{" > $OUT/catalog_sizes.jsonl; tail -2 $OUT/catalog_sizes.jsonl | cut -c1-300 ;;
prof)
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python $ROOT/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras > $OUT/bench_profiled_run.json 2> $OUT/prof.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_fits -- python $ROOT/bench_configs.py --configs fits,cfg4 > $OUT/prof_fits.jsonl 2> $OUT/prof_fits.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_catalog -- python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/prof_catalog.json 2> $OUT/prof_catalog.err
  cd $ROOT
  for d in prof prof_fits prof_catalog; do
    f=$(find $OUT/$d -name "*kernel_stats.csv" | head -1)
    [ -n "$f" ] && cp "$f" $OUT/kernel_stats_$d.csv
    find $OUT/$d -name "*.csv" -size +1M -delete
  done
  head -5 $OUT/kernel_stats_prof.csv; head -12 $OUT/kernel_stats_prof_fits.csv; head -16 $OUT/kernel_stats_prof_catalog.csv ;;
esac
done
