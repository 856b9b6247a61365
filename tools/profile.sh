#!/bin/bash
# Developer tool: rocprofv3 kernel-trace stats + separate PMC passes for bench.py (1 GPU).
# usage: tools/profile.sh <tag> [bench args...]     (run on the GPU box via gpurun)
set -u
TAG=${1:-r01}; shift || true
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
ARGS="--steps 5 --warmup 2 --no-cpu-baseline $*"
cd $PWD
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 bench.py $ARGS > $OUT/trace.log 2>&1
for pass in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_REQ_sum TCC_READ_sum" \
            "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD" \
            "SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_THREAD_CYCLES_VALU" \
            "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" \
            "TA_BUSY_avr TA_TA_BUSY_sum TCP_GATE_EN1_sum TCP_TA_TCP_STATE_READ_sum" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $pass --output-format csv -d $OUT/pmc_$name -o pmc -- python3 bench.py $ARGS > $OUT/pmc_$name.log 2>&1 || echo "pass failed: $pass" >> $OUT/errors.log
done
echo done > $OUT/done
