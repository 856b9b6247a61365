#!/bin/bash
# GPU box: march_phong_kernel (VV_PHONG_PAIR=0) against march_phong_pair_kernel (1: two x-adjacent slabs per block) over the Phong workloads.
run() { VV_BENCH_NO_EXTRA=1 timeout -k 10 300 python3 bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"; }
CFGS=("--phong --steps 20 --warmup 5" "--config c5 --steps 10 --warmup 3" "--phong --view b --steps 20 --warmup 5" "--config c2 --phong --steps 50 --warmup 10" "--config c1 --phong --steps 100 --warmup 10" \
      "--phong --volume brain --tf engine --steps 20 --warmup 5" "--phong --voxel u8 --steps 20 --warmup 5" "--phong --frame-of 8 --steps 10 --warmup 3" "--phong --size 512 --steps 20 --warmup 5")
if [ $# -gt 0 ]; then CFGS=("$@"); fi
for cfg in "${CFGS[@]}"; do
  line="$cfg |"
  for v in ${VERS:-0 1 0 1}; do line="$line pair=$v $(VV_PHONG_PAIR=$v run $cfg)"; done
  echo "$line"
done
