export TMPDIR=/tmp VV_BENCH_NO_EXTRA=1
OUT=$PWD/gpurun_out/pq_phong; rm -rf $OUT; mkdir -p $OUT
for pass in "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU" "GRBM_GUI_ACTIVE TA_BUSY_avr SQ_INSTS_SALU SQ_WAIT_ANY"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-30)
  rocprofv3 --pmc $pass --output-format csv -d $OUT/pmc_$name -o pmc -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --phong "$@" > $OUT/$name.log 2>&1
done
python3 tools/pmc_summary.py $OUT "march_phong"
