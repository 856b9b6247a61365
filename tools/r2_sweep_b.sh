#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sweep_kernel" > gpurun_out/sweep_tests.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/sweep_tests.log
for cfg in "3 3" ; do set -- $cfg
VV_SWEEP_NL=$1 VV_SWEEP_GROUP=$2 VV_SWEEP=1 VV_BENCH_NO_EXTRA=1 timeout -k 10 120 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench_sweep.log 2>&1
echo "nl=$1 group=$2 rc=$? $(grep -h kernel_ms gpurun_out/bench_sweep.log | sed -e 's/.*"executed_samples_per_frame": \([0-9]*\).*/\1/') $(grep -h kernel_ms gpurun_out/bench_sweep.log | sed -e 's/.*"ms_per_step": \([0-9.]*\).*"frac": \([0-9.]*\).*/ms_per_step \1 frac \2/')"
done
VV_SWEEP=1 timeout -k 10 200 python tools/sweep_stress.py 12 2>&1 | grep "^frame" | awk '{print $3, $5}' | sort | uniq -c | sort -rn | head -3
