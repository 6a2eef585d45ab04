#!/bin/bash
# compile ONE translation unit of csrc/ with the library's flags and print its resource table: tools/cc1.sh iso_fast_stretch_tree.hip
cd /root/repo/isochrones_amd/csrc || exit 1
out=/tmp/cc1_$(basename "$1" .hip)
time /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-fast-math -Wall -Wno-unused-function -Wno-bitwise-instead-of-logical -mllvm -disable-machine-licm ${CC1_ILP--mllvm -amdgpu-sched-strategy=max-ilp} -Rpass-analysis=kernel-resource-usage ${CC1_EXTRA} -c "$1" -o $out.o 2>$out.res
echo rc=$?
grep -v "remark\|^ *[0-9]* *|\|^ *|" $out.res | head -40
python resources.py $out.res | head -${2:-40}
