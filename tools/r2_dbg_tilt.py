import sys, os
sys.path.insert(0, "volume-viz_amd/python"); sys.path.insert(0, "tests")
os.environ["VV_SWEEP"] = "1"; os.environ["VV_SWEEP_VERBOSE"] = "1"
import numpy as np, volviz_amd as vv, oracle_lib as O
cam = vv.Camera.orbit(4.0, 1.2, -1.3)
with vv.Context(0) as ctx:
    vol = O.noise_u8(64, 64, 64, 7).astype(np.float32) / np.float32(255)
    tf = vv.transfer_preset(vv.TF_ENGINE)
    ctx.load_volume(vol, tf)
    o = dict(step=1 / 64, ert_mode=vv.ERT_REFERENCE, ert_threshold=0.9)
    got = ctx.render(150, 97, cam, options=vv.make_options(count_samples=True, **o), fill=0x11)
    print("counters", ctx.debug_counters().tolist())
