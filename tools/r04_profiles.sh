#!/bin/bash
# GPU box: the profile set of a round under the DRIVER's bench protocol (python3 bench.py --steps 20 --warmup 5: >= 300 launches of the headline
# kernel, so averages are the steady state the bench line reports) -> gpurun_out/<tag>/ ; copy the summaries into profiles/.
#   usage: tools/r04_profiles.sh <tag>
set -u
TAG=${1:-r04prof}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
python3 bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err || { echo "bench failed"; tail -3 $OUT/bench.err; exit 1; }
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/trace.json 2> $OUT/trace.err || echo "trace failed"
i=0
for pass in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum" \
            "TA_BUSY_avr TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" \
            "GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES" \
            "SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --pmc $pass --output-format csv -d $OUT/pmc_$i -o pmc -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/pmc_$i.json 2> $OUT/pmc_$i.err || echo "pmc pass $i failed"
done
python3 tools/pmc_summary.py $OUT "big::march_kernel<-1, 1, true, false, false" > $OUT/pmc_march_kernel.txt
python3 tools/pmc_summary.py $OUT "brick::march_kernel<-1, 1, true, false, false" > $OUT/pmc_march_kernel_viewb_bricked.txt
python3 tools/pmc_summary.py $OUT "big::march_phong_kernel<-1, 1, true, false>" > $OUT/pmc_march_phong_kernel.txt
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv 2>/dev/null
python3 tools/pmc_sub.py > $OUT/pmc_sub.log 2>&1 || echo "pmc_sub failed"
cp gpurun_out/pmc_sub.json $OUT/ 2>/dev/null
python3 tools/pmc_traffic.py > $OUT/pmc_traffic.log 2>&1 || echo "pmc_traffic failed"
cp gpurun_out/pmc_traffic.json $OUT/ 2>/dev/null
head -3 $OUT/pmc_march_kernel.txt; head -4 $OUT/kernel_stats.csv | cut -c1-200; tail -c 700 $OUT/bench.json
