# developer tool: VV_UNROLL=2 vs 3 with repeats for the bench arguments given
export VV_BENCH_NO_EXTRA=1
for rep in 1 2; do for u in ${UNROLLS:-2 3}; do
  echo -n "$* rep=$rep u=$u : "; env VV_UNROLL=$u timeout -k 10 120 python bench.py --steps 60 --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['kernel_ms_rank0'])"
done; done
