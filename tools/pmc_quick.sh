#!/bin/bash
# developer tool: a few PMC passes only (FETCH_SIZE, L2 requests, TA busy) for the current env
TAG=$1; shift
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pq_$TAG; mkdir -p $OUT
ARGS="--steps 3 --warmup 1 --no-cpu-baseline $*"
for pass in "FETCH_SIZE" "TCC_EA0_RDREQ_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TA_BUSY_avr TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" "GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-30)
  rocprofv3 --pmc $pass --output-format csv -d $OUT/pmc_$name -o pmc -- python3 bench.py $ARGS > $OUT/$name.log 2>&1
done
python3 tools/pmc_summary.py $OUT "march_"  | grep -v "n=  1 "
