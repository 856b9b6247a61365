#!/usr/bin/env python3
"""Developer tool (GPU box): what do the blocks of a frame cost when they have nothing to march?  C3's frame with the cube scaled down to a few pixels:
every block stages the table, sets its rays up, finds no sample and writes zeros -- the launch / set-up floor of march_kernel's grid (+ rad_kernel)."""
import os, sys
REPO = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(REPO, "volume-viz_amd", "python")); sys.path.insert(0, REPO)
import numpy as np, torch
import volviz_amd as vv
import bench

n, W, H = int(os.environ.get("N", 256)), 1920, 1080
dev = torch.device("cuda", 0)
ctx = vv.Context(0)
ts = torch.cuda.Stream(device=dev); torch.cuda.set_stream(ts); stream = vv.stream_handle(ts)
v8 = torch.empty(n ** 3, dtype=torch.uint8, device=dev)
ctx.generate_noise_device(v8.data_ptr(), n, n, n, 1, stream)
v32 = torch.empty(n ** 3, dtype=torch.float32, device=dev)
ctx.promote_device(v8.data_ptr(), v32.data_ptr(), n ** 3, stream)
ctx.load_volume_device(v32.data_ptr(), vv.VOXEL_F32, n, n, n, bench.ramp_tf(), stream)
torch.cuda.synchronize()
frame = torch.zeros(H * W, dtype=torch.int32, device=dev)
for name, cam in (("cube at 1 % scale (all blocks empty)", vv.Camera(scale=(0.01, 0.01, 0.01))), ("normal", vv.Camera())):
    for bw in ("32", "64"):
        os.environ["VV_BLOCK_W"] = bw; os.environ["VV_TILE_LOG2W"] = "5"; ctx.reread_env()
        o = vv.make_options(step=1.0 / 512)
        for _ in range(50):
            ctx.render_device(W, H, cam, frame.data_ptr(), options=o, stream=stream)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            ctx.render_device(W, H, cam, frame.data_ptr(), options=o, stream=stream)
        e1.record(); torch.cuda.synchronize()
        print(f"{name:40s} block_w {bw}: {e0.elapsed_time(e1) / 200 * 1000:.1f} us per frame")
