#!/bin/bash
# developer tool: run bench for a few env-var variants in one gpurun call
for v in "$@"; do
  echo "== $v"; env $v timeout -k 10 120 python bench.py --steps 10 --no-cpu-baseline ${BENCH_ARGS:-} 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['executed_samples_per_frame'])"
done
