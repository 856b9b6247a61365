#!/bin/bash
# local helper (dev container): gpurun with a wait for a free slot (exit code 3 = "no box or slot free right now, nothing charged").
#   usage: tools/gpu.sh <timeout-seconds> '<command>'
T=$1; shift
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout $T -- "$@"; rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 90
done
exit 3
