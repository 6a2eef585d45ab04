#!/bin/bash
# Pinned sampler soak (isochrone, 3 stars, 9 bands: every move replayed against the CPU oracle) over variant libraries
# built by tools/build_variant.py from the round-3 source state that made wrong accept / reject decisions (987096e).
#   tools/miscompile_hunt.sh SECONDS name1 name2 ...     (libraries variants/libs/libiso_hip_<name>.so)
# Results: gpurun_out/hunt/<name>.log (MISMATCH lines carry the run's configuration) and gpurun_out/hunt/summary.txt
secs=$1; shift
mkdir -p gpurun_out/hunt
for v in "$@"; do
  for seed in ${HUNT_SEEDS:-1}; do
    ISOCHRONES_AMD_LIB=$PWD/variants/libs/libiso_hip_$v.so SOAK_KIND=${HUNT_KIND:-iso} SOAK_NSTARS=${HUNT_NSTARS:-3} SOAK_NB=${HUNT_NB:-9} \
      SOAK_CATALOG_FRACTION=0 timeout $((secs + 120)) python tests/soak/soak_sampler.py $secs $seed > gpurun_out/hunt/$v.s$seed.log 2>&1
    echo "$v seed $seed rc=$? $(grep -c MISMATCH gpurun_out/hunt/$v.s$seed.log) mismatches; $(tail -1 gpurun_out/hunt/$v.s$seed.log)" >> gpurun_out/hunt/summary.txt
  done
done
cat gpurun_out/hunt/summary.txt
