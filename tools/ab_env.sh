#!/bin/bash
# GPU box: A/B of one environment setting against the default over the bench workloads that use march_kernel's 32 x 2 wave tiles.
#   usage: tools/ab_env.sh VV_BLOCK_W=64 [bench.py argument strings ...]      (default: the workloads that use march_kernel's 32 x 2 wave tiles)
SET=$1; shift
run() { VV_BENCH_NO_EXTRA=1 timeout -k 10 300 python3 bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"; }
CFGS=("--steps 20 --warmup 5" "--config c2 --steps 50 --warmup 10" "--config c1 --steps 100 --warmup 20" "--size 512 --steps 30 --warmup 5" "--voxel u8 --steps 20 --warmup 5" \
           "--frame-of 2 --steps 20 --warmup 5" "--frame-of 4 --steps 10 --warmup 3" "--frame-of 8 --steps 10 --warmup 3" "--volume brain --tf engine --steps 20 --warmup 5" \
           "--orbit 80,-90 --steps 20 --warmup 5" "--orbit 90,-80 --steps 20 --warmup 5" "--ert true --filter exact --steps 20 --warmup 5")
if [ $# -gt 0 ]; then CFGS=("$@"); fi
for cfg in "${CFGS[@]}"; do
  a=$(run $cfg); b=$(env $SET bash -c "$(declare -f run); run $cfg"); a2=$(run $cfg); b2=$(env $SET bash -c "$(declare -f run); run $cfg")
  echo "$cfg | default $a $a2 | $SET $b $b2"
done
