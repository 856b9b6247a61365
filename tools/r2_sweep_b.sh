#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
for cfg in "3 3 0" "3 3 6" "3 3 12" "3 3 24" "2 3 12" "3 2 12"; do set -- $cfg
VV_SWEEP_NL=$1 VV_SWEEP_GROUP=$2 VV_SWEEP_LEAD=$3 VV_SWEEP=1 VV_BENCH_NO_EXTRA=1 timeout -k 10 120 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench_sweep.log 2>&1
echo "nl=$1 group=$2 lead=$3 rc=$? $(grep -h kernel_ms gpurun_out/bench_sweep.log | sed -e 's/.*"executed_samples_per_frame": \([0-9]*\).*/\1/') $(grep -h kernel_ms gpurun_out/bench_sweep.log | sed -e 's/.*"ms_per_step": \([0-9.]*\).*"frac": \([0-9.]*\).*/ms_per_step \1 frac \2/')"
done
