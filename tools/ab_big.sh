export VV_BENCH_NO_EXTRA=1
for rep in 1 2; do for fb in 0 1; do
  echo -n "rep=$rep force_big=$fb : "; if [ $fb = 1 ]; then export VV_FORCE_BIG=1; else unset VV_FORCE_BIG; fi; timeout -k 10 120 python bench.py --steps 40 --warmup 3 --no-cpu-baseline "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"
done; done
