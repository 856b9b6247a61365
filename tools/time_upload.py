"""developer tool (1 GPU): throughput of the streamed volume upload (vv_load_volume_stream_*) from
pinned and from pageable host memory, with and without u8 -> f32 promotion on the device."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "volume-viz_amd", "python"))
import volviz_amd as vv

n = 1024
ctx = vv.Context(0)
tf = vv.transfer_preset(vv.TF_HEAD)
pin = torch.empty((n, n, n), dtype=torch.uint8).pin_memory()
pin.random_(0, 255)
pageable = pin.numpy().copy()
slab = 64
for name, host in (("pinned", pin.numpy()), ("pageable", pageable)):
    for vt, label in ((vv.VOXEL_U8, "u8 volume"), (vv.VOXEL_F32, "f32 volume, promoted on the device")):
        best = 1e9
        for _ in range(3):
            t = time.perf_counter()
            ctx.load_volume_streamed(((z, host[z:z + slab]) for z in range(0, n, slab)), vt, n, n, n, tf)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t)
        print(f"{name:9s} 1 GiB of u8 slices -> {label}: {best * 1e3:7.1f} ms = {2**30 / best / 1e9:5.1f} GB/s over PCIe")
