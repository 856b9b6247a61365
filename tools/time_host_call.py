import os, sys, time
import numpy as np, torch
REPO = os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."); sys.path.insert(0, os.path.join(REPO, "volume-viz_amd", "python")); sys.path.insert(0, REPO)
import volviz_amd as vv
from bench import ramp_tf
dev = torch.device("cuda", 0); ctx = vv.Context(0)
n = 256
v8 = torch.empty(n**3, dtype=torch.uint8, device=dev); ctx.generate_noise_device(v8.data_ptr(), n, n, n, 1)
ctx.load_volume_device(v8.data_ptr(), vv.VOXEL_U8, n, n, n, ramp_tf()); torch.cuda.synchronize()
frame = torch.zeros((64, 64, 4), dtype=torch.uint8, device=dev)
cam = vv.Camera(); o = vv.make_options(step=1/64, shard=(4, 8, 3))
ts = torch.cuda.Stream(device=dev); torch.cuda.set_stream(ts)
s = vv.stream_handle(ts)      # a caller stream: vv_render only enqueues
for _ in range(20): ctx.render_device(64, 64, cam, frame.data_ptr(), options=o, stream=s)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(2000): ctx.render_device(64, 64, cam, frame.data_ptr(), options=o, stream=s)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"host time per vv_render call (tiny frame): {(t1 - t) / 2000 * 1e6:.1f} us; incl. GPU drain {(t2 - t) / 2000 * 1e6:.1f} us")
