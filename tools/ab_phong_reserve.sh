#!/bin/bash
# A/B: resident blocks per CU of march_phong_kernel (VV_LDS_RESERVE_PHONG) on the big-volume workloads
for cfg in "--config c3" "--config c3 --view b" "--config c5 --steps 5" "--config c3 --frame-of 8"; do
  out="$cfg:"
  for res in 13000 22000 30000 40000 60000 100000; do
    ms=$(VV_LDS_RESERVE_PHONG=$res VV_BENCH_NO_EXTRA=1 timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --phong $cfg 2>/dev/null | tail -1 | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
    out="$out $res=$ms"
  done
  echo "$out"
done
