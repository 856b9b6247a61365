#!/bin/bash
# GPU box: EA bytes + time of the C3 headline under another build of the library (tools/run_with_lib.py) against this tree's.  usage: tools/ab_lib_pmc.sh <other_lib.so> <tag>
OLD=$1; TAG=${2:-ablib}; export TMPDIR=/tmp
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
for which in new old new old; do
  if [ $which = old ]; then cmd="python3 tools/run_with_lib.py $OLD bench.py"; else cmd="python3 bench.py"; fi
  ms=$(VV_BENCH_NO_EXTRA=1 timeout -k 10 400 $cmd --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
  echo "$which | $ms ms"
done
for which in new old; do
  if [ $which = old ]; then args="tools/run_with_lib.py $OLD bench.py"; else args="bench.py"; fi
  VV_BENCH_NO_EXTRA=1 VV_BENCH_SPINUP=30 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_$which -o pmc -- python3 $args --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2> $OUT/pmc_$which.err
  python3 - $OUT/pmc_$which $which <<'PY'
import csv, glob, sys, collections
acc=collections.defaultdict(list)
for f in glob.glob(sys.argv[1]+"/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "march_kernel" in r["Kernel_Name"] and "false, false, 3>" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(sys.argv[2], {k: round(sum(v)/len(v)) for k,v in acc.items()}, "EA GB", round(sum(acc["TCC_EA0_RDREQ_sum"])/max(1,len(acc["TCC_EA0_RDREQ_sum"]))*128/1e9,3))
PY
done
