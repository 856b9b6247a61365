#!/bin/bash
# A/B: one vs two slabs per Phong block (VV_PHONG_SPB) over volume sizes, views, voxel types and configurations
mkdir -p gpurun_out
run() { # label, args...
  local label=$1; shift
  local out=""
  for spb in 1 2; do
    local ms=$(VV_PHONG_SPB=$spb VV_BENCH_NO_EXTRA=1 timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --phong "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
    out="$out spb$spb=$ms"
  done
  echo "$label:$out"
}
run "c3 view a"
run "c3 view b (bricks)" --view b
run "c3 u8" --voxel u8
run "c3 u8 view b" --voxel u8 --view b
run "512^3" --size 512
run "256^3" --size 256
run "c2" --config c2
run "c2 view b" --config c2 --view b
run "c1" --config c1
run "c3 brain engine" --volume brain --tf engine
run "frame of 8" --frame-of 8
run "c5" --config c5 --steps 3 --warmup 1
