# developer tool: sweep of march_kernel's occupancy cap (VV_LDS_RESERVE) for the bench arguments given
export VV_BENCH_NO_EXTRA=1
for lr in ${RESERVES:-0 16000 28000 36000 49000 76000}; do
  echo -n "$* lds=$lr : "; env VV_LDS_RESERVE=$lr timeout -k 10 120 python bench.py --steps 40 --warmup 3 --no-cpu-baseline "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"
done
