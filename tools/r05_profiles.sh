#!/bin/bash
# GPU box: the profile set of round 5 under the DRIVER's bench protocol (python3 bench.py --steps 20 --warmup 5) -> gpurun_out/<tag>/ ; copy the summaries into profiles/.
# New against tools/r04_profiles.sh: the Phong frame and C5 get kernel-trace runs of their own (one kernel, one workload per CSV: each bench number is one line of one file),
# and the PMC passes carry TCC_EA0_RDREQ_DRAM beside TCC_EA0_RDREQ.
#   usage: tools/r05_profiles.sh <tag> [part ...]     parts: bench trace pmc phong pmcphong c5 sub traffic (default: all)
set -u
TAG=${1:-r05prof}; shift || true
PARTS=${*:-bench trace pmc phong pmcphong c5 sub traffic}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
has() { case " $PARTS " in *" $1 "*) return 0;; esac; return 1; }
stats() { cp "$(find $1 -name '*kernel_stats.csv' | head -1)" $2 2>/dev/null || echo "no kernel_stats in $1"; }
if has bench; then
  python3 bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err || { echo "bench failed"; tail -3 $OUT/bench.err; exit 1; }
fi
if has trace; then
  VV_BENCH_NO_EXTRA=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/trace.json 2> $OUT/trace.err || echo "trace failed"
  stats $OUT/trace $OUT/kernel_stats.csv
fi
if has phong; then
  VV_BENCH_NO_EXTRA=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_phong -o trace -- python3 bench.py --phong --steps 20 --warmup 5 --no-cpu-baseline > $OUT/trace_phong.json 2> $OUT/trace_phong.err || echo "phong trace failed"
  stats $OUT/trace_phong $OUT/kernel_stats_phong.csv
fi
if has c5; then
  VV_BENCH_NO_EXTRA=1 VV_BENCH_SPINUP=20 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_c5 -o trace -- python3 bench.py --config c5 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/trace_c5.json 2> $OUT/trace_c5.err || echo "c5 trace failed"
  stats $OUT/trace_c5 $OUT/kernel_stats_c5.csv
fi
if has pmc; then
  i=0
  for pass in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_HIT_sum TCC_MISS_sum" \
              "TA_BUSY_avr TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" \
              "GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES" \
              "SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
    i=$((i+1))
    rocprofv3 --pmc $pass --output-format csv -d $OUT/pmc_$i -o pmc -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/pmc_$i.json 2> $OUT/pmc_$i.err || echo "pmc pass $i failed"
  done
  python3 tools/pmc_summary.py $OUT "big::march_kernel<-1, 1, true, false, false" > $OUT/pmc_march_kernel.txt
  python3 tools/pmc_summary.py $OUT "brick::march_kernel<-1, 1, true, false, false" > $OUT/pmc_march_kernel_viewb_bricked.txt
  python3 tools/pmc_summary.py $OUT "zfast::march_kernel<-1, 1, true, false, false" > $OUT/pmc_march_kernel_side_zfast.txt
fi
if has pmcphong; then
  # the Phong frame alone (in the full bench run the same kernel name also marches C5's frames): its own passes, its own directory
  mkdir -p $OUT/phong; i=0
  for pass in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_HIT_sum TCC_MISS_sum" \
              "TA_BUSY_avr TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" \
              "GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES" \
              "SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
    i=$((i+1))
    VV_BENCH_NO_EXTRA=1 rocprofv3 --pmc $pass --output-format csv -d $OUT/phong/pmc_$i -o pmc -- python3 bench.py --phong --steps 20 --warmup 5 --no-cpu-baseline > $OUT/phong/pmc_$i.json 2> $OUT/phong/pmc_$i.err || echo "phong pmc pass $i failed"
  done
  python3 tools/pmc_summary.py $OUT/phong "march_phong_kernel<-1, 1, true, false>" > $OUT/pmc_march_phong_kernel.txt
fi
if has sub; then python3 tools/pmc_sub.py > $OUT/pmc_sub.log 2>&1 || echo "pmc_sub failed"; cp gpurun_out/pmc_sub.json $OUT/ 2>/dev/null; fi
if has traffic; then python3 tools/pmc_traffic.py > $OUT/pmc_traffic.log 2>&1 || echo "pmc_traffic failed"; cp gpurun_out/pmc_traffic.json $OUT/ 2>/dev/null; fi
head -3 $OUT/pmc_march_kernel.txt 2>/dev/null; head -4 $OUT/kernel_stats.csv 2>/dev/null | cut -c1-200; tail -c 400 $OUT/bench.json 2>/dev/null
