"""developer tool (1 GPU): per-rank march time of an N-rank frame when every rank takes one contiguous
range of slab rows (balanced by executed samples) instead of round-robin bands."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "volume-viz_amd", "python"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import volviz_amd as vv
from bench import FRAMES, ramp_tf

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = 1024
W, H, steps = FRAMES[N]
dev = torch.device("cuda", 0)
ctx = vv.Context(0)
v8 = torch.empty(n ** 3, dtype=torch.uint8, device=dev)
ctx.generate_noise_device(v8.data_ptr(), n, n, n, 0x9E3779B9)
v32 = torch.empty(n ** 3, dtype=torch.float32, device=dev)
ctx.promote_device(v8.data_ptr(), v32.data_ptr(), n ** 3)
ctx.load_volume_device(v32.data_ptr(), vv.VOXEL_F32, n, n, n, ramp_tf())
torch.cuda.synchronize()
del v8, v32
frame = torch.zeros((H + 256, W, 4), dtype=torch.uint8, device=dev)
cam = vv.Camera()
nby = (H + 13) // 14
# cost per slab row = executed samples
cost = []
for r in range(nby):
    ctx.render_device(W, H, cam, frame.data_ptr(), options=vv.make_options(step=1.0 / steps, slab_rows=(r, r + 1), count_samples=True))
    torch.cuda.synchronize()
    cost.append(ctx.last_sample_count())
cum = np.cumsum(cost); total = cum[-1]
cuts = [0] + [int(np.searchsorted(cum, total * k / N) + 1) for k in range(1, N)] + [nby]
ms = []
for k in range(N):
    o = vv.make_options(step=1.0 / steps, slab_rows=(cuts[k], cuts[k + 1]))
    t = []
    for _ in range(5):
        ctx.render_device(W, H, cam, frame.data_ptr(), options=o)
        torch.cuda.synchronize()
        t.append(ctx.last_frame_ms())
    ms.append(float(np.median(t)))
print(f"N={N} contiguous slab-row ranges {cuts}: ms per rank {['%.3f' % m for m in ms]} max/mean {max(ms) / np.mean(ms):.3f} sum {sum(ms):.3f}")
