#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sweep_kernel" > gpurun_out/sweep_tests.log 2>&1
echo "pytest rc=$?"; tail -4 gpurun_out/sweep_tests.log
for cfg in "2 4 2" "2 2 2" "2 8 2" "3 4 2" "4 4 2" "2 4 1" "2 1 3"; do set -- $cfg
VV_SWEEP_NL=$1 VV_SWEEP_GROUP=$2 VV_SWEEP_DEPTH=$3 VV_SWEEP=1 VV_BENCH_NO_EXTRA=1 timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench_sweep.log 2>&1
echo "nl=$1 group=$2 depth=$3 rc=$? $(grep -h kernel_ms gpurun_out/bench_sweep.log | sed -e 's/.*"ms_per_step": \([0-9.]*\).*"frac": \([0-9.]*\).*/ms_per_step \1 frac \2/')"
done
