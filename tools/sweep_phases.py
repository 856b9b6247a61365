"""Developer aid (GPU box, experimental library): the slab sweep (vv_sweep.hip) on the C3 frame for a list of shapes --
is the frame the gather kernel's (bit for bit), the frame time (plain launches, HIP events), and from one instrumented
frame: cycles of wave 0 per phase and block trip (copy issue / wait for own copies / barrier / march), bytes staged into
LDS, samples that fell outside the images (must be 0), blocks that left the ring.

    python3 tools/sweep_phases.py [--frames 20] [--size 1024] [--only 0,3]
"""
import argparse, json, os, sys
REPO = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(REPO, "volume-viz_amd", "python")); sys.path.insert(0, REPO)
import numpy as np, torch, volviz_amd as vv
import bench

SHAPES = [  # wx, wy, steps, ahead, blocks per CU (0 = the planner's choice)
    (2, 4, 1, 6, 0), (2, 4, 1, 6, 1), (2, 4, 2, 6, 1), (2, 4, 1, 1, 0),
    (2, 3, 1, 6, 0), (2, 2, 1, 6, 0), (2, 2, 1, 6, 3), (2, 2, 2, 6, 2), (1, 4, 1, 6, 0), (3, 4, 1, 6, 1), (3, 4, 2, 6, 1),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=20)
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    shapes = [SHAPES[int(i)] for i in args.only.split(",")] if args.only else SHAPES
    n, W, H, steps = args.size, 1920, 1080, args.size // 2
    ctx = vv.Context(0, lib_path=vv.LIB_X_PATH); dev = torch.device("cuda", 0); stream = torch.cuda.current_stream().cuda_stream
    v8 = torch.empty(n ** 3, dtype=torch.uint8, device=dev); ctx.generate_noise_device(v8.data_ptr(), n, n, n, 0x9E3779B9, stream)
    v32 = torch.empty(n ** 3, dtype=torch.float32, device=dev); ctx.promote_device(v8.data_ptr(), v32.data_ptr(), n ** 3, stream)
    ctx.load_volume_device(v32.data_ptr(), vv.VOXEL_F32, n, n, n, bench.ramp_tf(), stream); torch.cuda.synchronize(); del v8, v32
    frame = torch.zeros(H * W, dtype=torch.int32, device=dev)
    cam = vv.Camera()

    def timed(o):
        for _ in range(3):
            ctx.render_device(W, H, cam, frame.data_ptr(), options=o, stream=stream)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.frames):
            ctx.render_device(W, H, cam, frame.data_ptr(), options=o, stream=stream)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / args.frames

    for k in ("VV_SWEEP", "VV_SWEEP_WX", "VV_SWEEP_WY", "VV_SWEEP_STEPS", "VV_SWEEP_AHEAD", "VV_SWEEP_BLOCKS"):
        os.environ.pop(k, None)
    ctx.reread_env()
    plain = vv.make_options(step=1 / steps)
    ms_gather = timed(plain)
    want = frame.clone()
    ctx.render_device(W, H, cam, frame.data_ptr(), options=vv.make_options(step=1 / steps, count_samples=True), stream=stream); torch.cuda.synchronize()
    executed = int(ctx.debug_counters()[0])
    print(json.dumps({"gather_kernel_ms": round(ms_gather, 3), "executed_samples": executed}), flush=True)
    for wx, wy, st, ahead, blocks in shapes:
        env = {"VV_SWEEP": "1", "VV_SWEEP_WX": str(wx), "VV_SWEEP_WY": str(wy), "VV_SWEEP_STEPS": str(st), "VV_SWEEP_AHEAD": str(ahead)}
        os.environ.pop("VV_SWEEP_BLOCKS", None)
        if blocks:
            env["VV_SWEEP_BLOCKS"] = str(blocks)
        os.environ.update(env); ctx.reread_env()
        rec = {"tile_px": [32 * wx, 2 * wy], "steps_per_trip": st, "ahead": ahead, "blocks_per_cu": blocks or "auto"}
        try:
            frame.zero_()
            ms = timed(plain)
            rec["ms"] = round(ms, 3)
            rec["identical"] = bool(torch.equal(frame, want))
            frame.zero_()
            ctx.render_device(W, H, cam, frame.data_ptr(), options=vv.make_options(step=1 / steps, count_samples=True), stream=stream); torch.cuda.synchronize()
            c = [int(v) for v in ctx.debug_counters()]
            trips = max(c[12], 1)
            rec.update({"identical_instrumented": bool(torch.equal(frame, want)), "executed": c[0], "lane_use": round(c[0] / max(c[1], 1), 3),
                        "outside_images": c[4], "staged_GB": round(c[5] / 1e9, 3), "wave_trips": c[6], "gather_trips": c[13], "block_trips": c[12],
                        "cycles_per_block_trip": {"issue": round(c[8] / trips), "wait_own_copies": round(c[9] / trips), "barrier": round(c[10] / trips), "march": round(c[11] / trips)},
                        "err_counts": [(c[7] >> (16 * q)) & 0xFFFF for q in range(4)], "launch": ctx.last_launch()})
        except Exception as e:            # a shape the planner refuses, or a flagged frame
            rec["error"] = str(e)[:200]
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
