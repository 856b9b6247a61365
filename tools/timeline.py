#!/usr/bin/env python3
"""GPU box, experiment build only (make variant NAME=tl VFLAGS=-DVV_TIMELINE=1): start / end time of every block of one C3 frame of march_kernel.
   python3 tools/run_with_lib.py volume-viz_amd/lib_v/libvolviz_tl.so tools/timeline.py [--orbit th,ph] [--frames 3]
Prints how many blocks are resident over the frame, how long blocks live by where they are, and how far apart x-neighbours start."""
import argparse, os, sys, json
import numpy as np
REPO = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(REPO, "volume-viz_amd", "python")); sys.path.insert(0, REPO)
import torch
import volviz_amd as vv
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--orbit", default=""); ap.add_argument("--phong", action="store_true"); ap.add_argument("--frames", type=int, default=3); ap.add_argument("--size", type=int, default=1024)
ap.add_argument("--w", type=int, default=1920); ap.add_argument("--h", type=int, default=1080); ap.add_argument("--steps", type=int, default=512)
ap.add_argument("--volume", default="noise"); ap.add_argument("--tf", default="ramp"); ap.add_argument("--out", default="gpurun_out/timeline.npz")
a = ap.parse_args()
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
n, W, H, steps = a.size, a.w, a.h, a.steps
ctx = vv.Context(0)
ts = torch.cuda.Stream(device=dev); torch.cuda.set_stream(ts); stream = vv.stream_handle(ts)
v8 = torch.empty(n ** 3, dtype=torch.uint8, device=dev)
if a.volume == "noise": ctx.generate_noise_device(v8.data_ptr(), n, n, n, 0x9E3779B9, stream)
else: ctx.generate_default_brain_device(v8.data_ptr(), n, n, n, stream)
tf = {"ramp": bench.ramp_tf(), "head": vv.transfer_preset(vv.TF_HEAD), "engine": vv.transfer_preset(vv.TF_ENGINE)}[a.tf]
v32 = torch.empty(n ** 3, dtype=torch.float32, device=dev)
ctx.promote_device(v8.data_ptr(), v32.data_ptr(), n ** 3, stream)
ctx.load_volume_device(v32.data_ptr(), vv.VOXEL_F32, n, n, n, tf, stream)
torch.cuda.synchronize(); del v8, v32; torch.cuda.empty_cache()
cam = vv.Camera()
if a.orbit:
    th, ph = (float(v) for v in a.orbit.split(",")); cam = vv.Camera.orbit(4.0, np.radians(th), np.radians(ph))
opts = vv.make_options(step=1.0 / steps)
frame = torch.zeros((H, W, 4), dtype=torch.uint8, device=dev)
for _ in range(50): ctx.render_device(W, H, cam, frame.data_ptr(), options=opts, stream=stream, phong=a.phong)
torch.cuda.synchronize()
NB = 1 << 16
tl = torch.zeros(NB * 4, dtype=torch.int64, device=dev)
os.environ["VV_TIMELINE_PTR"] = str(tl.data_ptr())
res = []
for f in range(a.frames):
    tl.zero_(); torch.cuda.synchronize()
    ctx.render_device(W, H, cam, frame.data_ptr(), options=opts, stream=stream, phong=a.phong)
    torch.cuda.synchronize()
    res.append(tl.cpu().numpy().reshape(NB, 4).copy())
del os.environ["VV_TIMELINE_PTR"]
print("launch", ctx.debug_last_launch() if hasattr(ctx, "debug_last_launch") else "", "kernel ms", ctx.last_frame_ms())
T = res[-1]
used = T[:, 1] != 0
T = T[used]; idx = np.nonzero(used)[0]
t0, t1 = T[:, 0].astype(np.float64), T[:, 1].astype(np.float64)
base = t0.min(); t0 = (t0 - base) / 100.0; t1 = (t1 - base) / 100.0        # us (100 MHz)
strip, tile = (T[:, 2] >> 16).astype(int), (T[:, 2] & 0xffff).astype(int)
xcc, live, hwid = (T[:, 3] & 15).astype(int), (T[:, 3] >> 8) & 1, (T[:, 3] >> 16).astype(np.int64)
dur = t1 - t0
end = t1.max()
print(f"blocks {len(T)}  frame span {end:.1f} us   xcc == blockIdx % 8 for {np.mean(xcc == (idx & 7)) * 100:.1f} % of the blocks")
long_ = dur > 20.0
print(f"blocks living > 20 us: {long_.sum()}  their mean / median / p10 / p90 / max life: {dur[long_].mean():.1f} {np.median(dur[long_]):.1f} {np.percentile(dur[long_], 10):.1f} {np.percentile(dur[long_], 90):.1f} {dur[long_].max():.1f} us;  short blocks: mean {dur[~long_].mean():.2f} us, sum {dur[~long_].sum():.0f} us")
# resident marching blocks over time
grid = np.linspace(0, end, 41)
occ = [(int(((t0 <= g) & (t1 > g) & long_).sum()), int(((t0 <= g) & (t1 > g) & ~long_).sum())) for g in grid]
print("resident blocks (marching, short) at 40 points of the frame:")
print(" ".join(f"{m}" for m, s in occ))
busy = dur[long_].sum()
print(f"sum of marching-block lives {busy:.0f} us = {busy / end:.1f} blocks on average over the span;  at 512 slots the same work would take {busy / 512:.1f} us")
# when does occupancy fall below 90 % / 50 % of its plateau for good
plateau = np.median([m for m, s in occ[5:25]])
fine = np.linspace(0, end, 2001); occf = np.array([((t0 <= g) & (t1 > g) & long_).sum() for g in fine])
below90 = fine[np.nonzero(occf >= 0.9 * plateau)[0].max()]; below50 = fine[np.nonzero(occf >= 0.5 * plateau)[0].max()]
print(f"plateau {plateau:.0f} marching blocks; last time at >= 90 % of it: {below90:.1f} us, >= 50 %: {below50:.1f} us, end {end:.1f} us  -> tail {end - below90:.1f} us")
first = fine[np.nonzero(occf >= 0.9 * plateau)[0].min()]
print(f"ramp-up: 90 % of the plateau reached at {first:.1f} us")
# life by position: tile column classes and strip classes
for name, key in (("tile column", tile), ("strip / 16", strip // 16)):
    print(f"mean life of marching blocks by {name}:")
    print(" ".join(f"{k}:{dur[long_ & (key == k)].mean():.0f}" for k in sorted(set(key[long_]))))
# start gap between x-neighbours in a strip
gaps = []
order = np.lexsort((tile, strip))
for i, j in zip(order[:-1], order[1:]):
    if strip[i] == strip[j] and tile[j] == tile[i] + 1 and long_[i] and long_[j]: gaps.append(t0[j] - t0[i])
gaps = np.array(gaps)
print(f"start gap between x-neighbours (marching): median {np.median(gaps):.1f} us, p10 {np.percentile(gaps, 10):.1f}, p90 {np.percentile(gaps, 90):.1f}, mean |gap| {np.abs(gaps).mean():.1f}")
per_xcd_end = [t1[(xcc == k)].max() for k in range(8)]
print("last block end per XCD (us):", " ".join(f"{v:.0f}" for v in per_xcd_end))
os.makedirs(os.path.dirname(a.out), exist_ok=True)
np.savez_compressed(a.out, t0=t0, t1=t1, strip=strip, tile=tile, xcc=xcc, live=live, idx=idx, hwid=hwid)
