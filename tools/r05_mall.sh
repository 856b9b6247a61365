#!/bin/bash
# GPU box, round 5 item 1a: the re-read lag curve (tools/ubench/mall_reread.hip) plain and under counters, and the same counters on the headline kernel.
#   usage: tools/r05_mall.sh [tag]      -> gpurun_out/<tag>/
set -u
TAG=${1:-r05mall}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
B=tools/ubench/bin/mall_reread
[ -x $B ] || hipcc -O3 --offload-arch=gfx950 -o $B tools/ubench/mall_reread.hip
timeout -k 10 300 $B 3 > $OUT/mall_reread.txt 2>&1 || { echo "ubench failed"; tail -3 $OUT/mall_reread.txt; exit 1; }
# one dispatch per variant (reps = 0), counters per dispatch in launch order
timeout -k 10 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_ub -o pmc -- $B 0 > $OUT/mall_reread_pmc_run.txt 2>&1 || echo "ubench pmc failed"
timeout -k 10 300 rocprofv3 --pmc TCC_BUBBLE_sum TCC_EA0_RDREQ_32B_sum TCC_REQ_sum TCC_READ_sum --output-format csv -d $OUT/pmc_ub2 -o pmc -- $B 0 > $OUT/mall_reread_pmc2_run.txt 2>&1 || echo "ubench pmc2 failed"
python3 tools/r05_mall_join.py $OUT > $OUT/mall_reread_counters.txt 2>&1 || echo "join failed"
# the headline kernel: fabric requests against requests "destined for DRAM"
VV_BENCH_NO_EXTRA=1 timeout -k 10 400 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_march -o pmc -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/march_pmc.json 2> $OUT/march_pmc.err || echo "march pmc failed"
python3 tools/pmc_summary.py $OUT "march_kernel<-1, 1, true, false, false" > $OUT/pmc_march_kernel_dram.txt
cat $OUT/mall_reread.txt; cat $OUT/mall_reread_counters.txt; cat $OUT/pmc_march_kernel_dram.txt
