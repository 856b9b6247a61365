#!/bin/bash
# GPU box: A/B of this tree's library against another build (e.g. last round's) on one box.  usage: tools/ab_rounds.sh <other_lib.so> [configs...]
OLD=$1; shift
CFGS=("--config c5 --steps 10 --warmup 3" "--steps 20 --warmup 5" "--phong --steps 20 --warmup 5" "--view b --steps 20 --warmup 5")
if [ $# -gt 0 ]; then CFGS=("$@"); fi
for cfg in "${CFGS[@]}"; do
  for which in new old new old; do
    if [ $which = old ]; then cmd="python3 tools/run_with_lib.py $OLD bench.py"; else cmd="python3 bench.py"; fi
    ms=$(VV_BENCH_NO_EXTRA=1 timeout -k 10 400 $cmd $cfg --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
    echo "$cfg | $which | $ms"
  done
done
