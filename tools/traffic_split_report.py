#!/usr/bin/env python3
"""Joins tools/traffic_split.py's outputs: <dir>/time.jsonl (mode time), <dir>/pmc.jsonl (mode pmc: variant order) and the
rocprofv3 counter CSVs under <dir>/pmc_*/ -> a table (stdout).  Per variant the march dispatches are 1 instrumented +
F plain ones, in order; the plain ones are averaged."""
import collections, csv, glob, json, os, sys

root = sys.argv[1]
pats = (sys.argv[2] if len(sys.argv) > 2 else "march_,sweep_kernel").split(",")
tim = {}
if os.path.exists(os.path.join(root, "time.jsonl")):
    for l in open(os.path.join(root, "time.jsonl")):
        if l.startswith("{"):
            j = json.loads(l); tim[j["variant"]] = j
order = [json.loads(l) for l in open(os.path.join(root, "pmc.jsonl")) if l.startswith("{")]
counters = collections.OrderedDict()
for d in sorted(glob.glob(os.path.join(root, "pmc_*"))):
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        rows += [r for r in csv.DictReader(open(f)) if any(p in r["Kernel_Name"] for p in pats) and "rad_kernel" not in r["Kernel_Name"]]
    by = collections.defaultdict(list)
    for r in rows:
        by[r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    for cname, v in by.items():
        v.sort()
        vals = [x for _, x in v]
        per = []
        i = 0
        for o in order:
            g = vals[i:i + 1 + o["frames"]]; i += 1 + o["frames"]
            plain = g[1:]
            per.append(sum(plain) / len(plain) if plain else float("nan"))
        counters[cname] = per
        if i != len(vals):
            print(f"# warning: {cname}: {len(vals)} dispatches, expected {i}", file=sys.stderr)
names = list(counters)
print("variant | ms | samples M | algorithmic GB | EA read GB (RDREQ: 32B*32 + rest*128... see below) | x algorithmic | L2 hit | " + " | ".join(names))
for k, o in enumerate(order):
    t = tim.get(o["variant"], {})
    c = {n: counters[n][k] for n in names}
    ea = None
    if "TCC_EA0_RDREQ_sum" in c:
        r32 = c.get("TCC_EA0_RDREQ_32B_sum", 0.0)
        ea = (c["TCC_EA0_RDREQ_sum"] - r32) * 128.0 + r32 * 32.0 if "TCC_EA0_RDREQ_32B_sum" in c else c["TCC_EA0_RDREQ_sum"] * 128.0
    hit = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]) if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c else None
    alg = o["algorithmic_bytes"]
    print(f"{o['name']:48s} | {t.get('ms')} | {o['samples'] / 1e6:.1f} | {alg / 1e9:.3f} | {('%.3f' % (ea / 1e9)) if ea else None} | "
          f"{('%.3f' % (ea / alg)) if ea else None} | {('%.3f' % hit) if hit is not None else None} | " + " | ".join(f"{c[n]:.4g}" for n in names))
