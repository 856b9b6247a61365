"""developer tool (1 GPU): the fused ellipsoid generator at 1024^3 with 0, 1, 2, 4, 8 ellipsoids of drawDefaultBrain -- separates the fill from the per-ellipsoid work
-- and torch's own fill of the same buffer for comparison."""
import os, sys
import numpy as np, torch, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "volume-viz_amd", "python"))
import volviz_amd as vv
dev = torch.device("cuda", 0); ctx = vv.Context(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
v8 = torch.empty(n ** 3, dtype=torch.uint8, device=dev)
cen = [(0.25, .5, .5)] * 4 + [(0.75, .5, .5)] * 4
axs = [(.23, .30, .45), (.18, .27, .40), (.10, .23, .30), (.03, .20, .20)] * 2
col = [60, 80, 100, 120] * 2
stream = torch.cuda.current_stream().cuda_stream
def t(f, reps=10):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for m in (0, 1, 2, 4, 8):
    c = np.ascontiguousarray(cen[:m], np.float32).reshape(-1, 3); a = np.ascontiguousarray(axs[:m], np.float32).reshape(-1, 3); k = np.ascontiguousarray(col[:m], np.uint8)
    f = lambda: ctx._chk(ctx.lib.vv_generate_ellipsoids(ctx.h, v8.data_ptr(), 1, n, n, n, m, c.ctypes.data if m else None, a.ctypes.data if m else None, k.ctypes.data if m else None, stream))
    ms = t(f)
    print(f"n={n} ellipsoids={m}: {ms:.3f} ms = {n**3 / ms / 1e6:.0f} GB/s")
ms = t(lambda: v8.fill_(7))
print(f"torch fill_: {ms:.3f} ms = {n**3 / ms / 1e6:.0f} GB/s")
