"""developer tool (1 GPU): C3 from the side (camera on the x axis): the bricked copy (round 3's path for this view) against the z-fastest copy,
and the front view for reference; HIP-event time per frame and the frame's identity with the bricked path's."""
import os, sys
REPO = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(REPO, "volume-viz_amd", "python")); sys.path.insert(0, REPO)
import numpy as np, torch, volviz_amd as vv
import bench
n, W, H, steps = 1024, 1920, 1080, 512
ctx = vv.Context(0); dev = torch.device("cuda", 0); stream = torch.cuda.current_stream().cuda_stream
v8 = torch.empty(n ** 3, dtype=torch.uint8, device=dev); ctx.generate_noise_device(v8.data_ptr(), n, n, n, 0x9E3779B9, stream)
v32 = torch.empty(n ** 3, dtype=torch.float32, device=dev); ctx.promote_device(v8.data_ptr(), v32.data_ptr(), n ** 3, stream)
frame = torch.zeros(H * W, dtype=torch.int32, device=dev)
PHONG = len(sys.argv) > 1 and sys.argv[1] == "phong"
def timed(cam, frames=20):
    o = vv.make_options(step=1 / steps)
    for _ in range(30): ctx.render_device(W, H, cam, frame.data_ptr(), options=o, stream=stream, phong=PHONG)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(frames): ctx.render_device(W, H, cam, frame.data_ptr(), options=o, stream=stream, phong=PHONG)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / frames
cams = {"front (0, 0, -4)": vv.Camera(), "side (4, 0, 0)": vv.Camera(origin=(4.0, 0.0, 0.0)), "side (-4, 0, 0)": vv.Camera(origin=(-4.0, 0.0, 0.0)),
        "side, 8 degrees off": vv.Camera.orbit(4.0, np.pi / 2, np.radians(8.0)), "side, 20 degrees off (bricked by policy)": vv.Camera.orbit(4.0, np.pi / 2, np.radians(20.0))}
ref = {}
for zf in ("0", "auto"):
    if zf == "auto": os.environ.pop("VV_ZFAST", None)
    else: os.environ["VV_ZFAST"] = zf
    ctx.load_volume_device(v32.data_ptr(), vv.VOXEL_F32, n, n, n, bench.ramp_tf(), stream); torch.cuda.synchronize()
    for name, cam in cams.items():
        ms = timed(cam)
        lay = ctx.last_launch()["layout"]
        same = ""
        if zf == "0": ref[name] = frame.clone()
        else: same = "  identical to the VV_ZFAST=0 frame: %s" % bool(torch.equal(frame, ref[name]))
        print(f"VV_ZFAST={zf:4s} {name:42s} layout {lay}  {ms:.3f} ms{same}", flush=True)
