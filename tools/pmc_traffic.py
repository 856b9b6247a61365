#!/usr/bin/env python3
"""HBM-side bytes per launch of the shipped march kernels on the bench workload, from rocprofv3 PMC passes
(FETCH_SIZE and WRITE_SIZE in separate runs, counters only) -> gpurun_out/pmc_traffic.json, to be copied to
profiles/pmc_traffic.json.  The file carries the hash of volume-viz_amd/csrc it was measured with; bench.py reports
`roofline.traffic` only when that hash matches the tree.

FETCH_SIZE is in KB and on gfx950 counts a 128-byte request as 64 bytes (MI355X_MICROARCH.md, HBM): the factor is
calibrated in the same run on promote_kernel, which reads exactly n^3 bytes.  traffic = FETCH_SIZE*1024*factor +
WRITE_SIZE*1024.  The counter includes Infinity-Cache hits.          usage (GPU box): python3 tools/pmc_traffic.py
"""
import collections, csv, glob, json, os, subprocess, sys
REPO = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, REPO)
os.environ.setdefault("TMPDIR", "/tmp")
os.environ["VV_BENCH_NO_EXTRA"] = "1"
OUT = os.path.join(REPO, "gpurun_out", "pmc_traffic")
CONFIGS = {"c3-noise-ramp-a-n1": ["--view", "a"], "c3-noise-ramp-b-n1": ["--view", "b"], "c3-noise-ramp-a-phong-n1": ["--view", "a", "--phong"],
           "c3-noise-ramp-side-n1": ["--orbit", "90,180"], "c3-noise-ramp-a-u8-n1": ["--voxel", "u8"], "c3-noise-ramp-a-images-n1": ["--rays", "images"],
           "c5-noise-ramp-a-phong-n1": ["--config", "c5"]}
N = 1024


def run(tag, counter, extra):
    d = os.path.join(OUT, f"{tag}_{counter}")
    subprocess.run(["rocprofv3", "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "pmc", "--",
                    sys.executable, os.path.join(REPO, "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu-baseline"] + extra,
                   check=True, stdout=open(os.path.join(OUT, f"{tag}_{counter}.log"), "w"), stderr=subprocess.STDOUT, cwd=REPO)
    rows = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                rows[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return rows


def main():
    os.makedirs(OUT, exist_ok=True)
    import bench
    entries, detail = {}, {}
    only = sys.argv[1:]                      # optional: keys to (re-)measure; the others are kept from gpurun_out/pmc_traffic.json if its hash is this tree's
    prev = os.path.join(REPO, "gpurun_out", "pmc_traffic.json")
    if only and os.path.exists(prev):
        pj = json.load(open(prev))
        if pj.get("csrc_sha") == bench.csrc_sha():
            entries, detail = pj.get("entries", {}), pj.get("detail", {})
    for key, extra in CONFIGS.items():
        if only and key not in only:
            continue
        fetch, write = run(key, "FETCH_SIZE", extra), run(key, "WRITE_SIZE", extra)
        cal = [k for k in fetch if "promote_kernel" in k]
        # (C5 promotes 128 MiB slabs, not one n^3 volume: no calibration in that run, the factor of the C3 runs -- 1.9995 all round -- is used)
        factor = (N ** 3) / (sum(fetch[cal[0]]) / len(fetch[cal[0]]) * 1024.0) if (cal and "c5" not in extra) else 1.9995
        want = "march_phong_kernel" if ("--phong" in extra or "c5" in extra) else "march_kernel"
        # the timed launches: the uninstrumented instantiation is the one launched most often
        names = sorted((k for k in fetch if want in k), key=lambda k: -len(fetch[k]))
        if not names:
            continue
        k = names[0]
        f_kb = sum(fetch[k]) / len(fetch[k]); w_kb = sum(write[k]) / len(write[k]) if k in write else 0.0
        entries[key] = int(f_kb * 1024 * factor + w_kb * 1024)
        detail[key] = {"kernel": k[:120], "launches": len(fetch[k]), "FETCH_SIZE_KB": f_kb, "WRITE_SIZE_KB": w_kb, "fetch_factor": round(factor, 4)}
        print(key, entries[key], detail[key], flush=True)
    j = {"_comment": __doc__.strip().split("\n\n")[1].replace("\n", " "), "csrc_sha": bench.csrc_sha(), "entries": entries, "detail": detail}
    json.dump(j, open(os.path.join(REPO, "gpurun_out", "pmc_traffic.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
