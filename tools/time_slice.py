"""developer tool (1 GPU): time of the slice sampler writing into a device buffer."""
import os, sys, ctypes as C
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "volume-viz_amd", "python"))
import volviz_amd as vv

dev = torch.device("cuda", 0)
ctx = vv.Context(0)
n = 1024
v8 = torch.empty(n ** 3, dtype=torch.uint8, device=dev)
ctx.generate_noise_device(v8.data_ptr(), n, n, n, 3)
v32 = torch.empty(n ** 3, dtype=torch.float32, device=dev)
ctx.promote_device(v8.data_ptr(), v32.data_ptr(), n ** 3)
ctx.load_volume_device(v32.data_ptr(), vv.VOXEL_F32, n, n, n, vv.transfer_preset(vv.TF_HEAD))
torch.cuda.synchronize()
stream = torch.cuda.current_stream().cuda_stream
sc = (C.c_float * 3)(1.0, 1.0, 1.0)
for hw in (512, 1024, 2048, 4096):
    buf = torch.zeros(hw * hw, dtype=torch.float32, device=dev)
    for orient, name in ((vv.SAGITTAL, "sagittal (xy plane)"), (vv.CORONAL, "coronal (yz plane)"), (vv.HORIZONTAL, "horizontal (xz plane)")):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for it in range(12):
            if it == 2:
                e0.record()
            ctx._chk(ctx.lib.vv_slice(ctx.h, buf.data_ptr(), hw, hw, 0.3, 0.3, 0.3, orient, C.byref(sc), 0, vv.FILTER_TEX8, 1, stream))
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f"{hw}x{hw} {name:22s}: {ms * 1e3:8.1f} us  {hw * hw / ms / 1e6:7.1f} Gsamples/s")
