#!/bin/bash
# developer tool (GPU box): instruction counters of the slab sweep (experimental library) on the C3 frame, per launch.
#   bash tools/pmc_sweep.sh TAG "1,2"        # indices into tools/sweep_phases.py's SHAPES
TAG=$1; ONLY=${2:-1}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmcsw_$TAG; mkdir -p $OUT
for pass in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES" "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "FETCH_SIZE TCC_EA0_RDREQ_sum SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-30)
  rocprofv3 --pmc $pass --output-format csv -d $OUT/pmc_$name -o pmc -- python3 tools/sweep_phases.py --only $ONLY --frames 3 > $OUT/$name.log 2>&1 || { tail -5 $OUT/$name.log; exit 1; }
done
python3 - $OUT <<'PY'
import csv, glob, collections, os, sys
agg = collections.OrderedDict()
for f in sorted(glob.glob(os.path.join(sys.argv[1], 'pmc_*', '*counter_collection.csv'))):
    for r in csv.DictReader(open(f)):
        kn = r['Kernel_Name']
        if 'sweep_kernel' in kn or 'march_kernel' in kn:
            short = kn[kn.index('sweep_kernel'):kn.index('>') + 1] if 'sweep_kernel' in kn else 'march_kernel' + kn[kn.index('march_kernel') + 12:kn.index('>') + 1]
            agg.setdefault((short, r['Counter_Name']), []).append(float(r['Counter_Value']))
for (k, c), v in agg.items():
    print(f"{k:48s} {c:28s} n={len(v):3d} mean={sum(v)/len(v):.6g}")
PY
