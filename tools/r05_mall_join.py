#!/usr/bin/env python3
"""Joins the per-dispatch counters of tools/ubench/mall_reread (run with reps = 0: one dispatch per variant, launch order = print order)
with the variant table it printed.   usage: r05_mall_join.py gpurun_out/<tag>"""
import csv, glob, os, sys, collections
root = sys.argv[1]
rows = collections.OrderedDict()     # dispatch id -> {counter: value}
names = {}
for d in ("pmc_ub", "pmc_ub2"):
    for f in sorted(glob.glob(os.path.join(root, d, "**", "*counter_collection.csv"), recursive=True)):
        for r in csv.DictReader(open(f)):
            k = (r["Kernel_Name"].split("(")[0], int(r["Dispatch_Id"]))
            rows.setdefault(k, {})[r["Counter_Name"]] = float(r["Counter_Value"])
import re
variants = [l.rstrip("\n") for l in open(os.path.join(root, "mall_reread_pmc_run.txt")) if re.match(r"^(\s*\d+ |table)", l)]
for kern in ("reread", "loop_read"):
    ids = sorted(i for (k, i) in rows if k.startswith(kern))
    vs = [v for v in variants if (v.startswith("table") == (kern == "loop_read"))]
    print(f"# {kern}: variant | EA_RDREQ x128B GB | RDREQ_DRAM/RDREQ | L2 hit | 32B reqs | TCC_BUBBLE(128B reqs)")
    for n, i in enumerate(ids):
        c = rows[(next(k for (k, j) in rows if j == i and k.startswith(kern)), i)]
        rd, dr = c.get("TCC_EA0_RDREQ_sum", 0), c.get("TCC_EA0_RDREQ_DRAM_sum", 0)
        h, m = c.get("TCC_HIT_sum", 0), c.get("TCC_MISS_sum", 0)
        v = vs[n] if n < len(vs) else "?"
        print(f"{v:70s} | {rd * 128 / 1e9:7.3f} | {dr / rd if rd else 0:6.3f} | {h / (h + m) if h + m else 0:5.3f} | {c.get('TCC_EA0_RDREQ_32B_sum', -1):.0f} | {c.get('TCC_BUBBLE_sum', -1):.4g}")
