#!/bin/bash
# gpurun with retries while no GPU slot is free (exit code 3 = nothing charged).  usage: scripts/gpu_retry.sh <timeout> '<command>'
T=$1; shift
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout $T -- "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 60
done
exit 3
