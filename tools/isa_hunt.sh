#!/bin/bash
# Pinned sampler soak with the single-model persistent kernel (default priors, no asteroseismic terms) taken from hand-edited
# ISA (tools/isa_edit.py -> variants/hsaco/<name>.hsaco), launched by the round-3 tree's hunt hook (ISO_HSACO).
#   tools/isa_hunt.sh SECONDS name1 name2 ...
secs=$1; shift
mkdir -p gpurun_out/hunt
for v in "$@"; do
  ISO_HSACO=$PWD/variants/hsaco/$v.hsaco ISOCHRONES_AMD_LIB=$PWD/variants/libs/libiso_hip_modl.so SOAK_KIND=iso SOAK_NSTARS=3 SOAK_NB=9 \
    SOAK_CATALOG_FRACTION=0 SOAK_MODE=auto timeout $((secs + 120)) python tests/soak/soak_sampler.py $secs 1 > gpurun_out/hunt/isa_$v.log 2>&1
  python - "$v" <<'PY' >> gpurun_out/hunt/isa_summary.txt
import json, sys
v = sys.argv[1]
bad = tot = other = 0
for ln in open("gpurun_out/hunt/isa_%s.log" % v):
    if ln.startswith("MISMATCH") and '{"kind"' in ln:
        c = json.loads(ln[ln.index('{"kind"'):])
        if not c["priors"] and "nu_max" not in c["obs"]: bad += 1
        else: other += 1
last = [l for l in open("gpurun_out/hunt/isa_%s.log" % v)][-1].strip()
print("%s: %d wrong runs through the edited kernel (default priors, no asteroseismic terms); %d wrong runs of the other three compiled-in forms; %s" % (v, bad, other, last[:110]))
PY
done
cat gpurun_out/hunt/isa_summary.txt
