#!/bin/bash
# GPU box: several builds of the library against this tree's over a few bench workloads.  usage: tools/ab_libs.sh "<lib1.so> <lib2.so> ..." "<bench args>" ...
LIBS=$1; shift
for cfg in "$@"; do
  for rep in 1 2; do
    line="$cfg | base $(VV_BENCH_NO_EXTRA=1 timeout -k 10 300 python3 bench.py $cfg --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")"
    for l in $LIBS; do
      ms=$(VV_BENCH_NO_EXTRA=1 timeout -k 10 300 python3 tools/run_with_lib.py $l bench.py $cfg --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
      line="$line | $(basename $l .so | sed s/libvolviz_//) $ms"
    done
    echo "$line"
  done
done
