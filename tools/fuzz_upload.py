"""developer tool (1 GPU): the upload paths against each other -- a volume loaded whole (vv_load_volume_*), streamed in
random slabs in random order (vv_load_volume_stream_*, u8 slabs promoted on the device or f32 slabs, pageable and pinned
sources), from a .t3d file written by vv_t3d_write, and from a device pointer must render the same frame and the same
slices.  usage: python3 tools/fuzz_upload.py <first seed> <last seed + 1>"""
import os, sys, tempfile
import numpy as np
REPO = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(REPO, "tests")); sys.path.insert(0, os.path.join(REPO, "volume-viz_amd", "python"))
import torch
import volviz_amd as vv


def main():
    lo, hi = int(sys.argv[1]), int(sys.argv[2])
    ctx = vv.Context(0)
    bad = 0
    tmp = tempfile.mkdtemp()
    for seed in range(lo, hi):
        rng = np.random.default_rng(77000000 + seed)
        dims = tuple(int(v) for v in rng.choice([1, 2, 3, 7, 16, 31, 40, 64, 100, 256], size=3))       # nx, ny, nz
        if dims[0] * dims[1] * dims[2] > 4_000_000:
            dims = (dims[0], min(dims[1], 64), min(dims[2], 64))
        nx, ny, nz = dims
        v8 = rng.integers(0, 256, size=(nz, ny, nx), dtype=np.uint8)
        as_f32 = bool(rng.random() < 0.5)
        vol = (v8.astype(np.float32) / np.float32(255)) if as_f32 else v8
        tf = vv.transfer_preset(int(rng.choice([vv.TF_ENGINE, vv.TF_HEAD, vv.TF_MRI])))
        cam = vv.Camera.orbit(float(rng.uniform(1.5, 4.0)), float(rng.uniform(0.2, 2.9)), float(rng.uniform(-3, 3)))
        W, H = int(rng.integers(8, 90)), int(rng.integers(8, 70))
        phong = bool(rng.random() < 0.4)

        def shot():
            f = ctx.render(W, H, cam, phong=phong, options=vv.make_options(step=1 / 48), fill=5)
            s = ctx.slice(33, 29, 0.1, 0.2, 0.3, orientation=vv.HORIZONTAL, fill=-1.0)
            return f, s
        ctx.load_volume(vol, tf)
        ref = shot()
        results = {}
        # streamed, random slabs in random order; u8 slabs into an f32 volume are promoted on the device
        cuts = sorted(set([0, nz] + [int(v) for v in rng.integers(0, nz + 1, size=int(rng.integers(0, 6)))]))
        slabs = [(a, b) for a, b in zip(cuts[:-1], cuts[1:]) if b > a]
        order = rng.permutation(len(slabs))
        pinned = bool(rng.random() < 0.5)
        src_u8 = bool(rng.random() < 0.5) or not as_f32

        def gen():
            for k in order:
                a, b = slabs[k]
                arr = (v8 if src_u8 else vol)[a:b]
                if pinned:
                    t = torch.from_numpy(np.ascontiguousarray(arr)).pin_memory()
                    yield a, t.numpy()
                else:
                    yield a, arr
        ctx.load_volume_streamed(gen(), vv.VOXEL_F32 if as_f32 else vv.VOXEL_U8, nx, ny, nz, tf)
        results[f"streamed {len(slabs)} slabs pinned={pinned} u8-source={src_u8}"] = shot()
        # .t3d round trip (the reference's format stores u8)
        if not as_f32 or True:
            path = os.path.join(tmp, f"v{seed}.t3d")
            hdr = True                                # (the headerless form is the reference's fixed 128 x 256 x 256)
            rc = vv.load_library().vv_t3d_write(path.encode(), int(hdr), v8.ctypes.data, nx, ny, nz)
            assert rc == 0, rc
            ctx.load_volume_t3d(path, hdr, vv.VOXEL_F32 if as_f32 else vv.VOXEL_U8, tf)
            results[f"t3d header={hdr}"] = shot()
            os.remove(path)
        # device pointer
        dv = torch.from_numpy(vol).cuda()
        ctx.load_volume_device(dv.data_ptr(), vv.VOXEL_F32 if as_f32 else vv.VOXEL_U8, nx, ny, nz, tf)
        torch.cuda.synchronize()
        results["device pointer"] = shot()
        for what, (f, s) in results.items():
            if not (np.array_equal(f, ref[0]) and np.array_equal(s, ref[1])):
                bad += 1
                print("MISMATCH seed", seed, dims, "f32" if as_f32 else "u8", what, flush=True)
        if seed % 50 == 0:
            print("seed", seed, "mismatches so far", bad, flush=True)
        if bad > 8:
            break
    print(f"seeds {lo}..{hi - 1}: {bad} mismatches")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
