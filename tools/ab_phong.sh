# developer tool: sweep of the Phong kernel's occupancy cap (VV_LDS_RESERVE_PHONG) for the bench arguments given
export VV_BENCH_NO_EXTRA=1
for lr in ${RESERVES:-0 6000 13000 20000 28000 40000}; do
  echo -n "$* lds_phong=$lr : "; env VV_LDS_RESERVE_PHONG=$lr timeout -k 10 120 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --phong "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"
done
