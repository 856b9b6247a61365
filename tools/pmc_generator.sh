#!/bin/bash
# developer tool (GPU box): instruction / wait counters of the ellipsoid generator kernels at 1024^3 (tools/time_generator_parts.py), per launch.
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmcgen_$1; mkdir -p $OUT
for pass in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_ANY"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-30)
  rocprofv3 --pmc $pass --output-format csv -d $OUT/pmc_$name -o pmc -- python3 tools/time_generator_parts.py ${2:-1024} > $OUT/$name.log 2>&1 || { tail -5 $OUT/$name.log; exit 1; }
done
python3 - $OUT <<'PY'
import csv, glob, collections, os, sys
agg = collections.OrderedDict()
for f in sorted(glob.glob(os.path.join(sys.argv[1], 'pmc_*', '*counter_collection.csv'))):
    rows = [r for r in csv.DictReader(open(f)) if 'ellipsoid_rows_kernel' in r['Kernel_Name'] or 'ellipsoid_kernel' in r['Kernel_Name']]
    rows.sort(key=lambda r: int(r['Dispatch_Id']))
    # time_generator_parts.py: 13 launches per ellipsoid count (0, 1, 2, 4, 8), in that order
    for r in rows:
        agg.setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
for c, v in agg.items():
    per = len(v) // 5
    print(f"{c:28s} " + "  ".join(f"n={m}: {sum(v[i*per:(i+1)*per])/per:.4g}" for i, m in enumerate((0, 1, 2, 4, 8))))
PY
