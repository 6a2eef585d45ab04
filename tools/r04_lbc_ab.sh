#!/bin/bash
# A/B of the lane-per-sample BC gather's band limit (ISO_LANE_BC_MAX_BANDS) on single-star fits: default (4) vs 0 / 2 / 3
mkdir -p gpurun_out/r04b; out=gpurun_out/r04b/lane_bc_cap_ab.jsonl; rm -f $out
export SHAPES=track:1:1,track:1:2,track:1:3,track:1:4,track:1:5,iso:1:2,iso:1:3,iso:1:4
for W in 256 32; do
for v in default lbc0 lbc2 lbc3 default lbc0; do
  if [ $v = default ]; then unset ISOCHRONES_AMD_LIB; else export ISOCHRONES_AMD_LIB=$PWD/variants/libs/libiso_hip_$v.so; fi
  python tools/single_fit_shapes.py $W 2000 5 2>gpurun_out/r04b/sfs.err >> $out || tail -3 gpurun_out/r04b/sfs.err
done; done
python - <<PY
import json, collections
rows=[json.loads(l) for l in open("$out")]
t=collections.defaultdict(dict)
for r in rows:
    k=(r["walkers"],r["kind"],r["n_stars"],r["n_bands"]); lib=r["lib"].split("_")[-1].replace(".so","")
    t[k].setdefault(lib,[]).append((r["us_per_step"], r["lnprob_crc"]))
for k,v in t.items():
    print(k, {l:[round(x[0],2) for x in xs] for l,xs in v.items()}, "same chain" if len({x[1] for xs in v.values() for x in xs})==1 else "CHAINS DIFFER")
PY
