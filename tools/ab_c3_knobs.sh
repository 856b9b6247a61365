#!/bin/bash
# A/B of the gather kernel's launch knobs on the C3 frame (bench.py kernel time on a caller stream)
mkdir -p gpurun_out
for un in 2 3; do for res in 36000 49000 62000 76000 100000; do for tw in 5; do
  VV_UNROLL=$un VV_LDS_RESERVE=$res VV_BENCH_NO_EXTRA=1 timeout -k 10 120 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/ab_knobs.log 2>&1
  echo "unroll=$un reserve=$res $(grep -h kernel_ms gpurun_out/ab_knobs.log | sed -e 's/.*"ms_per_step": \([0-9.]*\).*"kernel_ms_rank0": \([0-9.]*\).*"frac": \([0-9.]*\).*/ms \1 kernel \2 frac \3/')"
done; done; done
for band in 0 1 2 4; do
  VV_XCD_BAND=$band VV_BENCH_NO_EXTRA=1 timeout -k 10 120 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/ab_knobs.log 2>&1
  echo "xcd_band=$band $(grep -h kernel_ms gpurun_out/ab_knobs.log | sed -e 's/.*"ms_per_step": \([0-9.]*\).*"kernel_ms_rank0": \([0-9.]*\).*"frac": \([0-9.]*\).*/ms \1 kernel \2 frac \3/')"
done
