"""Per-block timeline of the sweep kernel on the C3 frame (developer aid): how long tiles live, how busy CUs are."""
import os, sys
sys.path.insert(0, "volume-viz_amd/python"); sys.path.insert(0, ".")
os.environ["VV_SWEEP"] = "1"; os.environ["VV_SWEEP_TRACE"] = "1"
import numpy as np, torch, volviz_amd as vv
import bench
n, W, H, steps = 1024, 1920, 1080, 512
ctx = vv.Context(0); dev = torch.device("cuda", 0); stream = torch.cuda.current_stream().cuda_stream
v8 = torch.empty(n ** 3, dtype=torch.uint8, device=dev); ctx.generate_noise_device(v8.data_ptr(), n, n, n, 0x9E3779B9, stream)
v32 = torch.empty(n ** 3, dtype=torch.float32, device=dev); ctx.promote_device(v8.data_ptr(), v32.data_ptr(), n ** 3, stream)
ctx.load_volume_device(v32.data_ptr(), vv.VOXEL_F32, n, n, n, bench.ramp_tf(), stream); torch.cuda.synchronize(); del v8, v32
frame = torch.zeros(H * W, dtype=torch.int32, device=dev)
o = vv.make_options(step=1 / steps)
for _ in range(3):
    ctx.render_device(W, H, vv.Camera(), frame.data_ptr(), options=o, stream=stream)
torch.cuda.synchronize()
print("frame ms", ctx.last_frame_ms())
t = ctx.sweep_trace()
t = t[t[:, 7] == 1]
t0 = t[:, 0].min()
start, loop, end = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0, (t[:, 2] - t0) / 100.0     # microseconds
hw, xcc = (t[:, 3] >> np.uint64(32)).astype(np.int64), (t[:, 3] & np.uint64(0xffffffff)).astype(np.int64)
cu = ((xcc & 15) << 12) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 7) | ((hw >> 8) & 15)           # xcc, se, sh, cu
kmx = (t[:, 5] >> np.uint64(32)).astype(np.uint32).astype(np.int32).astype(np.int64); kmn = (t[:, 5] & np.uint64(0xffffffff)).astype(np.uint32).astype(np.int32).astype(np.int64)
work = kmx >= kmn
print("blocks", len(t), "with work", int(work.sum()), "distinct CUs", len(np.unique(cu)))
dur = end - start
print("span of the kernel (us)", end.max())
print("tile life (us): work mean %.1f median %.1f max %.1f | empty mean %.2f" % (dur[work].mean(), np.median(dur[work]), dur[work].max(), dur[~work].mean()))
print("prologue (start -> march) of working tiles: mean %.2f us" % (loop - start)[work].mean())
busy = {}
for c, d, w_ in zip(cu, dur, work):
    busy.setdefault(c, [0.0, 0]); busy[c][0] += d; busy[c][1] += int(w_)
b = np.array([v[0] for v in busy.values()]); k = np.array([v[1] for v in busy.values()])
print("per CU: busy us mean %.1f min %.1f max %.1f ; working tiles per CU mean %.2f min %d max %d" % (b.mean(), b.min(), b.max(), k.mean(), k.min(), k.max()))
sl = (kmx - kmn + 1)[work]
it = (t[:, 6] & np.uint64(0xffffffff)).astype(np.int64)[work]; st = (t[:, 6] >> np.uint64(32)).astype(np.int64)[work]
print("wave 0 of working tiles: steps mean %.0f, stall iterations mean %.0f; us per step %.3f" % (it.mean(), st.mean(), (dur[work] / np.maximum(it, 1)).mean()))
print("slices per working tile mean %.0f ; us per slice %.3f" % (sl.mean(), (dur[work] / sl).mean()))
order = np.argsort(end)[-5:]
print("last finishers: ", [(int(t[i, 4] >> np.uint64(32)), int(t[i, 4] & np.uint64(0xffffffff)), round(float(start[i]), 1), round(float(end[i]), 1)) for i in order])
