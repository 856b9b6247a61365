"""developer tool (1 GPU): time of the fused ellipsoid generator and the u8->f32 promotion at N^3."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "volume-viz_amd", "python"))
import volviz_amd as vv

dev = torch.device("cuda", 0)
ctx = vv.Context(0)
for n in (256, 512, 1024, 2048):
    v8 = torch.empty(n ** 3, dtype=torch.uint8, device=dev)
    v32 = torch.empty(n ** 3, dtype=torch.float32, device=dev) if n <= 1024 else None
    stream = torch.cuda.current_stream().cuda_stream
    for _ in range(2):
        ctx.generate_default_brain_device(v8.data_ptr(), n, n, n, stream)
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    e0.record()
    for _ in range(5):
        ctx.generate_default_brain_device(v8.data_ptr(), n, n, n, stream)
    e1.record()
    if v32 is not None:
        for _ in range(5):
            ctx.promote_device(v8.data_ptr(), v32.data_ptr(), n ** 3, stream)
    e2.record()
    torch.cuda.synchronize()
    g = e0.elapsed_time(e1) / 5
    p = e1.elapsed_time(e2) / 5
    print(f"N={n}: drawDefaultBrain (8 ellipsoids fused) {g:.3f} ms = {n**3 / g / 1e6:.0f} GB/s written"
          + (f"; promote {p:.3f} ms = {5 * n**3 / p / 1e6:.0f} GB/s moved" if v32 is not None else ""))
    del v8, v32
    torch.cuda.empty_cache()
