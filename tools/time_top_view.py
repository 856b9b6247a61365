import os, sys
sys.path.insert(0, "volume-viz_amd/python"); sys.path.insert(0, ".")
import numpy as np, torch, volviz_amd as vv
import bench
n, W, H, steps = 1024, 1920, 1080, 512
ctx = vv.Context(0); dev = torch.device("cuda", 0); stream = torch.cuda.current_stream().cuda_stream
v8 = torch.empty(n ** 3, dtype=torch.uint8, device=dev); ctx.generate_noise_device(v8.data_ptr(), n, n, n, 0x9E3779B9, stream)
v32 = torch.empty(n ** 3, dtype=torch.float32, device=dev); ctx.promote_device(v8.data_ptr(), v32.data_ptr(), n ** 3, stream)
ctx.load_volume_device(v32.data_ptr(), vv.VOXEL_F32, n, n, n, bench.ramp_tf(), stream); torch.cuda.synchronize()
frame = torch.zeros(H * W, dtype=torch.int32, device=dev)
def timed(cam, frames=20):
    o = vv.make_options(step=1 / steps)
    for _ in range(30): ctx.render_device(W, H, cam, frame.data_ptr(), options=o, stream=stream)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(frames): ctx.render_device(W, H, cam, frame.data_ptr(), options=o, stream=stream)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / frames
for name, cam in {"front": vv.Camera(), "top (0,4,0) up z": vv.Camera(origin=(0.0, 4.0, 0.0), up=(0.0, 0.0, 1.0)), "top up -z": vv.Camera(origin=(0.0, 4.0, 0.0), up=(0.0, 0.0, -1.0)),
                  "bottom": vv.Camera(origin=(0.0, -4.0, 0.0), up=(0.0, 0.0, 1.0)), "top, up x (screen x along z)": vv.Camera(origin=(0.0, 4.0, 0.0), up=(1.0, 0.0, 0.0)),
                  "orbit theta 10": vv.Camera.orbit(4.0, np.radians(10.0), -np.pi / 2), "front rolled 90 (up x)": vv.Camera(up=(1.0, 0.0, 0.0))}.items():
    ms = timed(cam); print(f"{name:34s} {ms:.3f} ms  {ctx.last_launch()}", flush=True)
