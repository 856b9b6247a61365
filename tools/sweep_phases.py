"""Developer aid (GPU box): where a sweep block's time goes -- cycles of wave 0 per phase (copy issue / wait for own copies / barrier / march),
summed over the blocks of one instrumented C3 frame, for a few tile shapes."""
import json, os, sys
sys.path.insert(0, "volume-viz_amd/python"); sys.path.insert(0, ".")
import numpy as np, torch, volviz_amd as vv
import bench
n, W, H, steps = 1024, 1920, 1080, 512
ctx = vv.Context(0); dev = torch.device("cuda", 0); stream = torch.cuda.current_stream().cuda_stream
v8 = torch.empty(n ** 3, dtype=torch.uint8, device=dev); ctx.generate_noise_device(v8.data_ptr(), n, n, n, 0x9E3779B9, stream)
v32 = torch.empty(n ** 3, dtype=torch.float32, device=dev); ctx.promote_device(v8.data_ptr(), v32.data_ptr(), n ** 3, stream)
ctx.load_volume_device(v32.data_ptr(), vv.VOXEL_F32, n, n, n, bench.ramp_tf(), stream); torch.cuda.synchronize(); del v8, v32
frame = torch.zeros(H * W, dtype=torch.int32, device=dev)
for env in ({"VV_SWEEP_WX": "2", "VV_SWEEP_WY": "4", "VV_SWEEP_STEPS": "2", "VV_SWEEP_AHEAD": "2"},
            {"VV_SWEEP_WX": "2", "VV_SWEEP_WY": "2", "VV_SWEEP_STEPS": "2", "VV_SWEEP_AHEAD": "4"},
            {"VV_SWEEP_WX": "2", "VV_SWEEP_WY": "2", "VV_SWEEP_STEPS": "1", "VV_SWEEP_AHEAD": "8"}):
    os.environ["VV_SWEEP"] = "1"; os.environ.update(env); ctx.reread_env()
    o = vv.make_options(step=1 / steps, count_samples=True)
    for _ in range(2):
        ctx.render_device(W, H, vv.Camera(), frame.data_ptr(), options=o, stream=stream)
    torch.cuda.synchronize()
    c = [int(v) for v in ctx.debug_counters()]
    trips = max(c[12], 1)
    print(json.dumps({"env": env, "instrumented_frame_ms": round(ctx.last_frame_ms(), 3), "block_trips": c[12],
                      "cycles_per_block_trip": {"issue": round(c[8] / trips), "wait_own_copies": round(c[9] / trips), "barrier": round(c[10] / trips), "march": round(c[11] / trips)},
                      "staged_GB": round(c[5] / 1e9, 2), "bails": c[7] >> 48}))
