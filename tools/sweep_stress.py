"""Repeats the C3 frame through the sweep kernel: per-frame time and checksum (developer aid: races show up as
outliers or differing checksums)."""
import os, sys, hashlib
sys.path.insert(0, "volume-viz_amd/python"); sys.path.insert(0, ".")
os.environ["VV_SWEEP"] = "1"
import numpy as np, torch, volviz_amd as vv
import bench
n, W, H, steps = 1024, 1920, 1080, 512
ctx = vv.Context(0); dev = torch.device("cuda", 0); stream = torch.cuda.current_stream().cuda_stream
v8 = torch.empty(n ** 3, dtype=torch.uint8, device=dev); ctx.generate_noise_device(v8.data_ptr(), n, n, n, 0x9E3779B9, stream)
v32 = torch.empty(n ** 3, dtype=torch.float32, device=dev); ctx.promote_device(v8.data_ptr(), v32.data_ptr(), n ** 3, stream)
ctx.load_volume_device(v32.data_ptr(), vv.VOXEL_F32, n, n, n, bench.ramp_tf(), stream); torch.cuda.synchronize(); del v8, v32
frame = torch.zeros(H * W, dtype=torch.int32, device=dev)
o = vv.make_options(step=1 / steps)
io = vv.make_options(step=1 / steps, count_samples=True)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for k in range(N):
    inst = (k % 5) == 4
    frame.zero_()
    ctx.render_device(W, H, vv.Camera(), frame.data_ptr(), options=io if inst else o, stream=stream)
    torch.cuda.synchronize()
    h = hashlib.sha256(frame.cpu().numpy().tobytes()).hexdigest()[:12]
    extra = ""
    if inst:
        c = ctx.debug_counters()
        extra = f" samples {int(c[0])} misses {int(c[4])} err {int(c[7]):#x}"
    print(f"frame {k}: {ctx.last_frame_ms():9.3f} ms  {h}{extra}", flush=True)
