#!/usr/bin/env python3
"""Developer tool (GPU box): C3 view a marched from (i) analytic rays, (ii) analytic rays rounded to RGBA8 like the FBO does, (iii) two first-pass
images at three times the render size with the camera as hint, (iv) the same without a hint through a synchronous call: ms per frame and the launch chosen.
Separates what the 8-bit end points cost from what fetching them costs."""
import os, sys
REPO = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(REPO, "volume-viz_amd", "python")); sys.path.insert(0, REPO)
import numpy as np, torch
import volviz_amd as vv, bench
n, W, H = 1024, 1920, 1080
dev = torch.device("cuda", 0); ctx = vv.Context(0)
ts = torch.cuda.Stream(device=dev); torch.cuda.set_stream(ts); stream = vv.stream_handle(ts)
v8 = torch.empty(n ** 3, dtype=torch.uint8, device=dev); ctx.generate_noise_device(v8.data_ptr(), n, n, n, 0x9E3779B9, stream)
v32 = torch.empty(n ** 3, dtype=torch.float32, device=dev); ctx.promote_device(v8.data_ptr(), v32.data_ptr(), n ** 3, stream)
ctx.load_volume_device(v32.data_ptr(), vv.VOXEL_F32, n, n, n, bench.ramp_tf(), stream); torch.cuda.synchronize(); del v8, v32
frame = torch.zeros(H * W, dtype=torch.int32, device=dev)
opts = vv.make_options(step=1 / 512)
for view in ("a", "b"):
    cam = vv.Camera() if view == "a" else vv.Camera.orbit(4.0, np.pi / 3, np.pi / 5)
    iw, ih = 3 * W, 3 * H
    df = torch.empty(ih * iw * 4, dtype=torch.uint8, device=dev); db = torch.empty_like(df)
    ctx.first_pass_device(iw, ih, cam, df.data_ptr(), db.data_ptr(), stream); torch.cuda.synchronize()
    cases = [("analytic", vv.analytic_rays(cam), stream), ("analytic, end points rounded to RGBA8", vv.analytic_rays(cam, quantize8=True), stream),
             ("first-pass images 3x, hinted", vv.device_image_rays(df.data_ptr(), db.data_ptr(), iw, ih, hint=cam), stream),
             ("first-pass images 1x, hinted", None, stream)]
    d1f = torch.empty(H * W * 4, dtype=torch.uint8, device=dev); d1b = torch.empty_like(d1f)
    ctx.first_pass_device(W, H, cam, d1f.data_ptr(), d1b.data_ptr(), stream); torch.cuda.synchronize()
    cases[3] = ("first-pass images 1x, hinted", vv.device_image_rays(d1f.data_ptr(), d1b.data_ptr(), W, H, hint=cam), stream)
    for rep in range(2):
        for name, rs, st in cases:
            for _ in range(100):
                ctx.render_device(W, H, cam, frame.data_ptr(), rays=rs, options=opts, stream=st)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30):
                ctx.render_device(W, H, cam, frame.data_ptr(), rays=rs, options=opts, stream=st)
            e1.record(); torch.cuda.synchronize()
            print(f"view {view} | {name:42s} | {e0.elapsed_time(e1) / 30:.4f} ms | {ctx.last_launch()}", flush=True)
    del df, db
