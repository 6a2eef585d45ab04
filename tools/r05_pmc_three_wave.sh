# Counters of the register-capped catalog kernel on the reference-shape catalog (10^4 stars x 300 walkers), launched with three
# waves per workgroup (default) and with four (ISOCHRONES_AMD_DENSE_THREADS=256): one rocprofv3 --pmc pass each.
#   bash tools/r05_pmc_three_wave.sh      (GPU box; summary -> gpurun_out/r05/pmc_three_wave.txt)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for threads in 192 256; do
  export ISOCHRONES_AMD_DENSE_THREADS=$threads
  rm -rf /tmp/pmc3w_$threads
  timeout 600 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc3w_$threads -- \
      python $ROOT/tools/ab_kernels.py --cases cfg5ref > /tmp/pmc3w_$threads.log 2>&1
done
unset ISOCHRONES_AMD_DENSE_THREADS
python - <<PY | tee $OUT/pmc_three_wave.txt
import csv, glob, collections
for threads in (192, 256):
    f = glob.glob("/tmp/pmc3w_%d/**/*counter_collection.csv" % threads, recursive=True)
    if not f:
        print(threads, "no counter file"); continue
    rows = list(csv.DictReader(open(f[0])))
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.Counter()
    for r in rows:
        k = r["Kernel_Name"]
        if "k_stretch_persist" not in k or ", true, false, false, false>" not in k: continue
        if int(r.get("Workgroup_Size", r.get("Workgroup_Size_X", "0")) or 0) not in (192, 256): continue
        if int(r["Grid_Size"]) < 1000 * 192: continue          # the 10^4-star launches
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
        if r["Counter_Name"] == "SQ_WAVES": n[k] += 1
    for k, c in acc.items():
        L = max(n[k], 1)
        print("threads per workgroup %d: %s, %d launches" % (threads, k.split("(")[0][-60:], L))
        for name in ("SQ_WAVES", "GRBM_GUI_ACTIVE", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_ANY"):
            print("   %-22s %.4g per launch" % (name, c.get(name, 0) / L))

PY
