#!/usr/bin/env python3
"""Developer tool: summarise rocprofv3 PMC csv directories for kernels matching a substring."""
import csv, glob, collections, os, sys
root, pat = sys.argv[1], sys.argv[2]
agg = collections.OrderedDict()
for f in sorted(glob.glob(os.path.join(root, 'pmc_*', '*counter_collection.csv'))):
    for r in csv.DictReader(open(f)):
        if pat in r['Kernel_Name']:
            agg.setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
for k, v in agg.items():
    print(f"{k:32s} n={len(v):3d} mean={sum(v)/len(v):.5g}")
