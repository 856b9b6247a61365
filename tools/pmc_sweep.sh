#!/bin/bash
# developer tool: SQ / LDS / TCC counters of the sweep kernel on the C3 frame (separate passes, counters only)
TAG=${1:-sweep}; shift
export TMPDIR=/tmp
export VV_SWEEP=1 VV_BENCH_NO_EXTRA=1
OUT=$PWD/gpurun_out/pq_$TAG; mkdir -p $OUT
ARGS="--steps 3 --warmup 1 --no-cpu-baseline $*"
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
            "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
            "SQ_INST_CYCLES_VMEM SQ_INSTS_FLAT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_INSTS_BRANCH SQ_WAVES GRBM_GUI_ACTIVE" \
            "FETCH_SIZE" "TCC_EA0_RDREQ_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-30)
  rocprofv3 --pmc $pass --output-format csv -d $OUT/pmc_$name -o pmc -- python3 bench.py $ARGS > $OUT/$name.log 2>&1
done
python3 tools/pmc_summary.py $OUT "sweep_kernel" | grep -v "n=  1 "
