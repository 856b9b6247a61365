# developer tool: the same frame on volumes of neighbouring edge lengths (pitch aliasing check)
export VV_BENCH_NO_EXTRA=1
for s in ${SIZES:-1000 1016 1024 1032}; do
  echo -n "$* size=$s : "; timeout -k 10 120 python bench.py --steps 40 --warmup 3 --no-cpu-baseline --size $s "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], round(d['roofline']['algorithmic_bytes_per_launch']/1e9/d['ms_per_step'],3))"
done
