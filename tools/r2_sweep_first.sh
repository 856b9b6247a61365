#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
tools/ubench/bin/lds_pairs > gpurun_out/lds_pairs.txt 2>&1 || echo "ubench failed"
cat gpurun_out/lds_pairs.txt
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sweep_kernel" > gpurun_out/sweep_tests.log 2>&1
echo "pytest rc=$?"; tail -12 gpurun_out/sweep_tests.log
for nl in 2 4; do
VV_SWEEP_NL=$nl VV_SWEEP=1 VV_STATS=1 VV_BENCH_NO_EXTRA=1 timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench_sweep_nl$nl.log 2>&1
echo "bench sweep nl=$nl rc=$?"; grep -h "stats\|kernel_ms" gpurun_out/bench_sweep_nl$nl.log | sed -e 's/.*"ms_per_step": \([0-9.]*\).*"frac": \([0-9.]*\).*/ms_per_step \1 frac \2/'
done
