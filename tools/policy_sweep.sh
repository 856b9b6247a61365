#!/bin/bash
# GPU box: re-sweep of the launch policy (blocks per CU x samples per trip x block shape) over the bench workloads with the current build.
#   usage: tools/policy_sweep.sh > gpurun_out/policy_sweep.txt
run() { VV_BENCH_NO_EXTRA=1 timeout -k 10 200 python3 bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"; }
CFGS=("--steps 20 --warmup 5" "--config c2 --steps 50 --warmup 10" "--size 512 --steps 30 --warmup 5" "--voxel u8 --steps 20 --warmup 5" "--frame-of 2 --steps 20 --warmup 5" "--frame-of 4 --steps 10 --warmup 3" "--frame-of 8 --steps 10 --warmup 3" "--view b --steps 20 --warmup 5" "--view b --frame-of 8 --steps 10 --warmup 3" "--view b --config c2 --steps 50 --warmup 10")
for cfg in "${CFGS[@]}"; do
  echo "== $cfg : policy $(run $cfg)"
  for res in 36000 49000 76000; do for bw in 32 64 16; do for u in 2 3; do
    case "$cfg" in *"view b"*) [ $bw = 64 ] && continue;; *) [ $bw = 16 ] && continue;; esac
    echo "   reserve $res block_w $bw unroll $u : $(VV_LDS_RESERVE=$res VV_BLOCK_W=$bw VV_UNROLL=$u run $cfg)"
  done; done; done
done
