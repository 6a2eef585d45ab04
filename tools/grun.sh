#!/bin/bash
# gpurun with retries while the pod's GPU slots are busy (exit code 3 = nothing charged): tools/grun.sh <timeout-seconds> '<command>'
t=$1; shift
for attempt in $(seq 1 40); do
    /usr/local/graft/bin/gpurun --timeout "$t" -- "$@"
    rc=$?
    if [ $rc -ne 3 ]; then exit $rc; fi
    sleep 60
done
exit 3
