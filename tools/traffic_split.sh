#!/bin/bash
# GPU box: times + three PMC passes of tools/traffic_split.py -> gpurun_out/<tag>/ ; usage: tools/traffic_split.sh <tag> [traffic_split.py args]
set -u
TAG=${1:-ts}; shift || true
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
python3 tools/traffic_split.py --mode time "$@" > $OUT/time.jsonl 2> $OUT/time.err || { echo "time run failed"; tail -5 $OUT/time.err; exit 1; }
i=0
for pass in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum" "TA_BUSY_avr TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" "GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES"; do
  i=$((i+1))
  rocprofv3 --pmc $pass --output-format csv -d $OUT/pmc_$i -o pmc -- python3 tools/traffic_split.py --mode pmc "$@" > $OUT/pmc.jsonl 2> $OUT/pmc_$i.err || { echo "pmc pass $i failed"; tail -5 $OUT/pmc_$i.err; }
  echo "pass $i done"
done
python3 tools/traffic_split_report.py $OUT > $OUT/report.txt 2> $OUT/report.err
cat $OUT/report.txt
