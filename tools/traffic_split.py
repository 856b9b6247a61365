#!/usr/bin/env python3
"""Developer tool (GPU box): where do the gather kernel's bytes beyond the algorithmic ones come from?

One process sets up the C3 workload once and renders it under a list of variants (VV_* knobs re-read
between variants with vv_reread_env, cropped slab-row ranges through vv_render_options), each variant:
one instrumented frame (executed samples + algorithmic bytes of exactly that frame) and F plain frames.

    python3 tools/traffic_split.py --mode time                    # HIP-event times per variant
    rocprofv3 --pmc <counters> ... -- python3 tools/traffic_split.py --mode pmc
                                                                    # F = 3 frames per variant; the k-th group of 3
                                                                    # uninstrumented march dispatches belongs to variant k
    python3 tools/traffic_split_report.py gpurun_out/ts            # joins both into a table

Variants are chosen to separate (i) the y-halo between strips that sit on different XCDs (xcd_band),
(ii) tiles of one strip that do not meet in L2 because they start at different times (a crop that fits the
resident block slots starts every tile together: "one round"), (iii) occupancy / L1 effects (lds_reserve, unroll).
"""
import argparse
import json
import os
import sys

REPO = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(REPO, "volume-viz_amd", "python"))
sys.path.insert(0, REPO)
import numpy as np
import torch
import volviz_amd as vv

# (every name here is read by vv_knobs::read(); a variant that sets another VV_* name is refused below)
KNOBS = ("VV_XCD_BAND", "VV_LDS_RESERVE", "VV_UNROLL", "VV_TILE_LOG2W", "VV_LDS_RESERVE_PHONG", "VV_BRICKED", "VV_ZPAIR", "VV_BLOCK_W", "VV_TAIL", "VV_ZFAST", "VV_FORCE_BIG")

# name, env, slab_rows (None = whole frame), extra
VARIANTS = [
    ("base", {}, None),
    ("xcd_band=2", {"VV_XCD_BAND": "2"}, None),
    ("xcd_band=4", {"VV_XCD_BAND": "4"}, None),
    ("xcd_band=0 (raster)", {"VV_XCD_BAND": "0"}, None),
    ("1 block/CU", {"VV_LDS_RESERVE": "155000"}, None),
    ("3 blocks/CU", {"VV_LDS_RESERVE": "49000"}, None),
    ("4 blocks/CU", {"VV_LDS_RESERVE": "36000"}, None),
    ("unroll=2", {"VV_UNROLL": "2"}, None),
    ("tile 16x4", {"VV_TILE_LOG2W": "4"}, None),
    ("rows 36-40 (7 strips: <1 per XCD, one round)", {}, (36, 40)),
    ("rows 34-43 (16 strips: one round)", {}, (34, 43)),
    ("rows 34-43, xcd_band=2", {"VV_XCD_BAND": "2"}, (34, 43)),
    ("rows 30-48 (32 strips: two rounds)", {}, (30, 48)),
    ("rows 20-57 (66 strips: four rounds)", {}, (20, 57)),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", default="time", choices=["time", "pmc"])
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--only", default="", help="comma-separated variant indices")
    ap.add_argument("--phong", action="store_true")
    ap.add_argument("--view", default="a")
    ap.add_argument("--variants", default="", help="python file defining VARIANTS (overrides the built-in list)")
    args = ap.parse_args()
    variants = VARIANTS
    if args.variants:
        ns = {}
        exec(open(args.variants).read(), ns)
        variants = ns["VARIANTS"]
    if args.only:
        variants = [variants[int(i)] for i in args.only.split(",")]
    for name, env, _ in variants:
        unknown = [k for k in env if k.startswith("VV_") and k not in KNOBS]
        if unknown:
            sys.exit(f"variant {name!r} sets {unknown}: not a knob the library reads (it would silently measure the default)")
    import bench
    n, W, H, steps = args.size, 1920, 1080, 512
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    ctx = vv.Context(0)
    ts = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(ts)
    stream = vv.stream_handle(ts)
    v8 = torch.empty(n ** 3, dtype=torch.uint8, device=dev)
    ctx.generate_noise_device(v8.data_ptr(), n, n, n, 0x9E3779B9, stream)
    v32 = torch.empty(n ** 3, dtype=torch.float32, device=dev)
    ctx.promote_device(v8.data_ptr(), v32.data_ptr(), n ** 3, stream)
    ctx.load_volume_device(v32.data_ptr(), vv.VOXEL_F32, n, n, n, bench.ramp_tf(), stream)
    torch.cuda.synchronize()
    del v8, v32
    torch.cuda.empty_cache()
    cam = vv.Camera() if args.view == "a" else vv.Camera.orbit(4.0, np.pi / 3, np.pi / 5)
    frame = torch.zeros(H * W, dtype=torch.int32, device=dev)
    nb = (n + 7) // 8
    F = 3 if args.mode == "pmc" else 30
    if args.mode == "time":           # reach the steady state first (bench.py does the same)
        o = vv.make_options(step=1.0 / steps)
        for _ in range(300):
            ctx.render_device(W, H, cam, frame.data_ptr(), options=o, stream=stream, phong=args.phong)
        torch.cuda.synchronize()
    for vi, (name, env, rows) in enumerate(variants):
        for k in KNOBS:
            os.environ.pop(k, None)
        os.environ.update(env)
        ctx.reread_env()
        kw = dict(step=1.0 / steps)
        if rows:
            kw["slab_rows"] = rows
        bitmap = torch.zeros((nb ** 3 + 31) // 32, dtype=torch.int32, device=dev)
        ctx.render_device(W, H, cam, frame.data_ptr(), options=vv.make_options(count_samples=True, touched_bricks=bitmap.data_ptr(), **kw),
                          stream=stream, phong=args.phong)
        torch.cuda.synchronize()
        samples = ctx.last_sample_count()
        cnt = [int(v) for v in ctx.debug_counters()]
        nbricks = int(np.unpackbits(bitmap.cpu().numpy().view(np.uint8)).sum())
        nrows = (rows[1] - rows[0]) * 14 if rows else H
        alg = nbricks * 512 * 4 + 4 * W * nrows + 4096
        o = vv.make_options(**kw)
        ms = None
        if args.mode == "time":
            for _ in range(10):
                ctx.render_device(W, H, cam, frame.data_ptr(), options=o, stream=stream, phong=args.phong)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(F):
                ctx.render_device(W, H, cam, frame.data_ptr(), options=o, stream=stream, phong=args.phong)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / F
        else:
            for _ in range(F):
                ctx.render_device(W, H, cam, frame.data_ptr(), options=o, stream=stream, phong=args.phong)
            torch.cuda.synchronize()
        print(json.dumps({"variant": vi, "name": name, "env": env, "rows": rows, "frames": F, "samples": samples,
                          "algorithmic_bytes": alg, "ms": None if ms is None else round(ms, 4), "counters": cnt}), flush=True)


if __name__ == "__main__":
    main()
