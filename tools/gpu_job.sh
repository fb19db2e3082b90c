#!/bin/bash
# Submit one gpurun call, retrying while no slot is free (exit 3: nothing charged).  usage: tools/gpu_job.sh <timeout> '<command>'
T=$1; shift
for i in 1 2 3 4 5 6 7 8 9 10; do
  /usr/local/graft/bin/gpurun --timeout $T -- "$@"
  rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 90
done
exit 3
