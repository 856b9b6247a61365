#!/bin/bash
# GPU box: this tree's 4x4x4 f32 bricks against another build's (e.g. lib_v/libvolviz_z8.so: 4x4x8, -DVV_BRICK_ZLOG2=3) on views off the memory axis:
# ms per frame (bench.py, two runs each) and EA bytes of the march kernel (one PMC pass each).   usage: tools/ab_brick_shape.sh <other_lib.so> <tag>
OLD=$1; TAG=${2:-brickab}; export TMPDIR=/tmp VV_BENCH_NO_EXTRA=1
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
ms() { tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['algorithmic_bytes_per_launch'])"; }
for cfg in "--view b --steps 20 --warmup 5" "--config c2 --view b --steps 50 --warmup 10" "--orbit 45,45 --steps 20 --warmup 5" "--orbit 90,-45 --steps 20 --warmup 5" "--orbit 20,10 --steps 20 --warmup 5" "--view b --voxel u8 --steps 20 --warmup 5"; do
  line="$cfg |"
  for which in new old new old; do
    if [ $which = old ]; then cmd="python3 tools/run_with_lib.py $OLD bench.py"; else cmd="python3 bench.py"; fi
    line="$line $which $(timeout -k 10 400 $cmd $cfg --no-cpu-baseline 2>/dev/null | ms) |"
  done
  echo "$line"
done
for cfg in "--view b" "--orbit 45,45"; do
  for which in new old; do
    if [ $which = old ]; then args="tools/run_with_lib.py $OLD bench.py"; else args="bench.py"; fi
    d=$OUT/pmc_${which}_$(echo $cfg | tr -d ' ,-')
    VV_BENCH_SPINUP=30 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $d -o pmc -- python3 $args $cfg --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2> $d.err
    python3 - $d "$which $cfg" <<'PY'
import csv, glob, sys, collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1]+"/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "march_kernel" in r["Kernel_Name"]: acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
k=max(acc, key=lambda k: len(acc[k]["TCC_EA0_RDREQ_sum"]))
v=acc[k]["TCC_EA0_RDREQ_sum"]
print(sys.argv[2], "| EA GB", round(sum(v)/len(v)*128/1e9,3), "| launches", len(v), "|", k[:60])
PY
  done
done
