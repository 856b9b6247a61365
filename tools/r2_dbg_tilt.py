import sys, os
sys.path.insert(0, "volume-viz_amd/python"); sys.path.insert(0, "tests")
os.environ["VV_SWEEP"] = "1"
import numpy as np, volviz_amd as vv, oracle_lib as O
cam = vv.Camera.orbit(4.0, 1.2, -1.3)
with vv.Context(0) as ctx:
    vol = O.draw_default_brain(128, 128, 128).astype(np.float32) / np.float32(255)
    tf = vv.transfer_preset(vv.TF_ENGINE)
    ctx.load_volume(vol, tf)
    o = dict(step=1 / 128, ert_mode=vv.ERT_REFERENCE, ert_threshold=0.9)
    got = ctx.render(320, 200, cam, options=vv.make_options(count_samples=True, **o), fill=0x11)
    print("counters", ctx.debug_counters().tolist())
    want, n = O.render(vol, tf, 320, 200, cam, options=vv.make_options(**o), fill=0x11)
    print("equal", np.array_equal(got, want), ctx.last_sample_count(), n)
