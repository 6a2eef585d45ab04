# A/B of variant libraries on the GPU box: bash tools/r05_ab.sh <cases> <lib name>...   ("default" = the in-tree library)
# one JSON line per library under gpurun_out/r05/ab_<cases>.jsonl
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05
mkdir -p $OUT
cases=$1; shift
tag=$(echo $cases | tr ',/' '__')
for rep in 1 2; do
for name in "$@"; do
  if [ "$name" = default ]; then unset ISOCHRONES_AMD_LIB; else export ISOCHRONES_AMD_LIB=$ROOT/variants/libs/libiso_hip_$name.so; fi
  python tools/ab_kernels.py --cases $cases --label $name 2>/dev/null | grep '^{' >> $OUT/ab_$tag.jsonl
done
done
unset ISOCHRONES_AMD_LIB
python - <<PY
import json
for l in open("$OUT/ab_$tag.jsonl"):
    d = json.loads(l)
    row = [d["label"]]
    for k, v in d.items():
        if isinstance(v, dict):
            if "ms_min" in v: row.append("%s %.2f us" % (k, v["ms_min"] * 1e3))
            elif "us_per_step" in v: row.append("%s %.3f us/step dig %.6f" % (k, v["us_per_step"], v["chain_digest"]))
            elif "wall_s_min" in v: row.append("%s %.2f ms %s dig %.9g" % (k, v["wall_s_min"] * 1e3, v["breakdown_last"], v["rows_digest"]))
    print(" | ".join(row))
PY
