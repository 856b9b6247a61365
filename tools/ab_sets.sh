#!/bin/bash
# GPU box: several environment settings against the default on bench workloads.  usage: tools/ab_sets.sh <tag> "<bench args>" -- "<SET1>" "<SET2>" ...
TAG=$1; shift
CFGS=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do CFGS+=("$1"); shift; done; shift
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
run() { VV_BENCH_NO_EXTRA=1 timeout -k 10 300 python3 bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"; }
for cfg in "${CFGS[@]}"; do
  line="$cfg | default $(run $cfg) $(run $cfg)"
  for S in "$@"; do line="$line | $S $(env $S bash -c "$(declare -f run); run $cfg") $(env $S bash -c "$(declare -f run); run $cfg")"; done
  echo "$line" | tee -a $OUT/ab.txt
done
