"""developer tool (1 GPU): per-rank march time and executed samples of an N-rank frame, rendered
one shard at a time, for different band heights -> how well the round-robin bands balance."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "volume-viz_amd", "python"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import volviz_amd as vv
from bench import FRAMES, ramp_tf

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
bands = [int(b) for b in sys.argv[2].split(",")] if len(sys.argv) > 2 else [4, 2, 1]
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
for band in bands:
    ms, smp = [], []
    for r in range(N):
        o = vv.make_options(step=1.0 / steps, shard=(band, N, r), count_samples=True)
        ctx.render_device(W, H, cam, frame.data_ptr(), options=o)
        torch.cuda.synchronize()
        smp.append(ctx.last_sample_count())
        o = vv.make_options(step=1.0 / steps, shard=(band, N, r))
        t = []
        for _ in range(5):
            ctx.render_device(W, H, cam, frame.data_ptr(), options=o)
            torch.cuda.synchronize()
            t.append(ctx.last_frame_ms())
        ms.append(float(np.median(t)))
    print(f"N={N} band={band} slab rows: ms per rank {['%.3f' % m for m in ms]}  max/mean {max(ms) / np.mean(ms):.3f}  "
          f"samples max/mean {max(smp) / np.mean(smp):.3f}  sum_ms {sum(ms):.3f}")
