#!/bin/bash
# GPU box: settings against the default, interleaved and repeated (box-state drift averages out).  usage: tools/ab_rep.sh <tag> <reps> "<bench args>" ... -- "<SET1>" ...
TAG=$1; REPS=$2; shift 2
CFGS=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do CFGS+=("$1"); shift; done; shift
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
run() { VV_BENCH_NO_EXTRA=1 timeout -k 10 300 python3 bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"; }
for cfg in "${CFGS[@]}"; do
  declare -A acc; for S in "default" "$@"; do acc["$S"]=""; done
  for r in $(seq 1 $REPS); do
    acc["default"]="${acc["default"]} $(run $cfg)"
    for S in "$@"; do acc["$S"]="${acc["$S"]} $(env $S bash -c "$(declare -f run); run $cfg")"; done
  done
  line="$cfg"
  for S in "default" "$@"; do m=$(python3 -c "import sys; v=sorted(float(x) for x in sys.argv[1:]); print('%.4f' % v[len(v)//2])" ${acc["$S"]}); line="$line | $S med $m (${acc["$S"]} )"; done
  echo "$line" | tee -a $OUT/ab.txt
done
