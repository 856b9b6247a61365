"""developer tool (1 GPU): the seeded random parity sweep of tests/test_gpu_parity.py over many more seeds,
cycling the layout knobs.  Prints the first mismatch, if any."""
import os, sys
import numpy as np
REPO = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(REPO, "tests")); sys.path.insert(0, os.path.join(REPO, "volume-viz_amd", "python"))
import importlib.util
import oracle_lib as O
import volviz_amd as vv
spec = importlib.util.spec_from_file_location("t", os.path.join(REPO, "tests", "test_gpu_parity.py"))
t = importlib.util.module_from_spec(spec); spec.loader.exec_module(t)

lo, hi = int(sys.argv[1]), int(sys.argv[2])
ctx = vv.Context(0)
envs = [{}, {"VV_BRICKED": "1"}, {"VV_ZPAIR": "1"}, {"VV_FORCE_BIG": "1"}, {"VV_PITCH_FORCE": "1"}, {"VV_BRICKED": "1", "VV_PITCH_FORCE": "1"}]
bad = 0
for seed in range(lo, hi):
    env = envs[seed % len(envs)]
    for k in ("VV_BRICKED", "VV_ZPAIR", "VV_FORCE_BIG", "VV_PITCH_FORCE"):
        os.environ.pop(k, None)
    os.environ.update(env)
    vol, tf, W, H, cam, sp, phong, o = t._random_case(seed)
    ctx.load_volume(vol, tf)
    opts = vv.make_options(**o)
    got = ctx.render(W, H, cam, slice=sp, phong=phong, options=opts, fill=0x3C)
    n_got = ctx.last_sample_count()
    want, n = O.render(vol, tf, W, H, cam, slice=sp, phong=phong, options=opts, fill=0x3C)
    if not np.array_equal(got, want) or n_got != n:
        bad += 1
        print("MISMATCH seed", seed, env, vol.shape, vol.dtype, W, H, phong, o, int((got != want).any(axis=-1).sum()), n_got, n)
        if bad > 5:
            break
print(f"seeds {lo}..{hi - 1}: {bad} mismatches")
