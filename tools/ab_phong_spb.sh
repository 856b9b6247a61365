#!/bin/bash
# A/B: slabs per Phong block x compact cache refresh x occupancy cap on the C3 + Phong frame
mkdir -p gpurun_out
for compact in 0 1; do for spb in 1 2; do for res in default 13000 30000 50000; do
  if [ $res = default ]; then unset VV_LDS_RESERVE_PHONG; else export VV_LDS_RESERVE_PHONG=$res; fi
  VV_PHONG_COMPACT=$compact VV_PHONG_SPB=$spb VV_BENCH_NO_EXTRA=1 timeout -k 10 120 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --phong "$@" > gpurun_out/ab_phong.log 2>&1
  echo "compact=$compact spb=$spb reserve=$res $(grep -h kernel_ms gpurun_out/ab_phong.log | sed -e 's/.*"ms_per_step": \([0-9.]*\).*"frac": \([0-9.]*\).*/ms \1 frac \2/')"
done; done; done
