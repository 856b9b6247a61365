#!/usr/bin/env python3
"""Issue fractions of the cache-resident configuration C2 ("not HBM-bound; report Msamples/s + LDS/VALU fraction") from
rocprofv3 PMC passes -> gpurun_out/pmc_sub.json, to be copied to profiles/pmc_sub.json.  bench.py's `c2` sub-record quotes it when the file's
hash of volume-viz_amd/csrc matches the tree.

valu_issue_fraction = SQ_ACTIVE_INST_VALU / (32 x GRBM_GUI_ACTIVE): SQ_ACTIVE_INST_* count quad-cycles a SIMD spends issuing that class, summed over the
1024 SIMDs; GRBM_GUI_ACTIVE counts busy cycles summed over the 8 XCDs, so 1024 x (GRBM / 8) / 4 = 32 x GRBM quad-cycles were available.
                                                                                   usage (GPU box): python3 tools/pmc_sub.py"""
import collections, csv, glob, json, os, subprocess, sys
REPO = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, REPO)
os.environ.setdefault("TMPDIR", "/tmp")
os.environ["VV_BENCH_NO_EXTRA"] = "1"
OUT = os.path.join(REPO, "gpurun_out", "pmc_sub")
PASSES = ["SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS", "GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD",
          "TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TA_BUSY_avr"]


def run(tag, counters, extra):
    d = os.path.join(OUT, tag)
    subprocess.run(["rocprofv3", "--pmc", *counters.split(), "--output-format", "csv", "-d", d, "-o", "pmc", "--",
                    sys.executable, os.path.join(REPO, "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu-baseline"] + extra,
                   check=True, stdout=open(os.path.join(OUT, tag + ".log"), "w"), stderr=subprocess.STDOUT, cwd=REPO)
    rows = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return rows


def main():
    os.makedirs(OUT, exist_ok=True)
    import bench
    entries, detail = {}, {}
    for key, extra, want in (("c2", ["--config", "c2", "--tf", "head"], "march_kernel"), ("c2-phong", ["--config", "c2", "--tf", "head", "--phong"], "march_phong_kernel"),
                             ("u8", ["--voxel", "u8"], "march_kernel")):
        acc = {}
        name = None
        for i, p in enumerate(PASSES):
            rows = run(f"{key}_{i}", p, extra)
            names = sorted((k for k in rows if want in k), key=lambda k: -max(len(v) for v in rows[k].values()))
            if not names:
                continue
            name = names[0]                       # the uninstrumented instantiation is the one launched most often
            for c, v in rows[name].items():
                acc[c] = sum(v) / len(v)
        if not acc or "GRBM_GUI_ACTIVE" not in acc:
            continue
        avail = 32.0 * acc["GRBM_GUI_ACTIVE"]
        entries[key] = {"valu_issue_fraction": round(acc.get("SQ_ACTIVE_INST_VALU", 0.0) / avail, 4), "lds_issue_fraction": round(acc.get("SQ_ACTIVE_INST_LDS", 0.0) / avail, 4),
                        "wave_wait_fraction": round(acc.get("SQ_WAIT_ANY", 0.0) / max(acc.get("SQ_WAVE_CYCLES", 1.0), 1.0), 4),
                        "ea_read_bytes": int(acc.get("TCC_EA0_RDREQ_sum", 0.0) * 128)}
        detail[key] = {"kernel": name[:140], "counters": {k: round(v, 1) for k, v in acc.items()}}
        print(key, entries[key], flush=True)
    j = {"_comment": __doc__.strip().split("\n\n")[1].replace("\n", " "), "csrc_sha": bench.csrc_sha(), "entries": entries, "detail": detail}
    json.dump(j, open(os.path.join(REPO, "gpurun_out", "pmc_sub.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
