"""Developer tool: print where the GPU frame and the oracle frame differ for one case."""
import sys, os
import numpy as np
REPO = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(REPO, "tests")); sys.path.insert(0, os.path.join(REPO, "volume-viz_amd", "python"))
import oracle_lib as O, volviz_amd as vv
from golden.make_fixtures import PLANE_POINT, PLANE_NORMAL

def cam(tag, scale=(1,1,1)):
    if tag == "a": return vv.Camera(scale=scale)
    if tag == "b": return vv.Camera.orbit(4.0, np.pi/3, np.pi/5, scale=scale)
    return vv.Camera.orbit(3.2, 2.0, -1.1, scale=scale)

def report(name, got, want):
    d = np.abs(got.astype(int) - want.astype(int)).max(axis=-1)
    print(f"{name}: exact {np.mean(d==0):.5f}  >1: {np.sum(d>1)}  max {d.max()}")
    ys, xs = np.nonzero(d > 1)
    for y, x in list(zip(ys, xs))[:8]:
        print("   px", (x, y), "got", got[y, x], "want", want[y, x])

ctx = vv.Context(0)
cases = [
 (170,170,(64,64,64),np.uint8,vv.TF_ENGINE,vv.SLICE_NONE,True,"b",vv.FILTER_EXACT),
 (170,170,(64,64,64),np.uint8,vv.TF_ENGINE,vv.SLICE_NONE,False,"b",vv.FILTER_EXACT),
 (170,170,(64,64,64),np.uint8,vv.TF_ENGINE,vv.SLICE_NONE,False,"b",vv.FILTER_TEX8),
 (97,131,(40,33,57),np.uint8,vv.TF_ENGINE,vv.SLICE_PLANE_CUT,False,"c",vv.FILTER_EXACT),
 (97,131,(40,33,57),np.uint8,vv.TF_ENGINE,vv.SLICE_NONE,False,"c",vv.FILTER_EXACT),
 (97,131,(40,33,57),np.float32,vv.TF_ENGINE,vv.SLICE_NONE,False,"c",vv.FILTER_EXACT),
]
for W,H,dims,dt,tfp,st,phong,ct,filt in cases:
    vol = O.draw_default_brain(*dims)
    if dt == np.float32: vol = vol.astype(np.float32)/np.float32(255)
    tf = vv.transfer_preset(tfp); ctx.load_volume(vol, tf)
    sp = vv.make_slice_params(st, PLANE_POINT, PLANE_NORMAL)
    for ert in (vv.ERT_REFERENCE, vv.ERT_TRUE, 2):
        opts = vv.make_options(filter=filt, count_samples=True, ert_mode=min(ert,1), ert_threshold=(2.0 if ert==2 else 0.0))
        got = ctx.render(W,H,cam(ct),slice=sp,phong=phong,options=opts)
        want,n = O.render(vol,tf,W,H,cam(ct),slice=sp,phong=phong,options=opts)
        report(f"{W}x{H} {dims} {np.dtype(dt).name} tf{tfp} s{st} p{int(phong)} {ct} f{filt} ert{ert} n={n}/{ctx.last_sample_count()}", got, want)
# colour tf + scale
rng = np.random.default_rng(11); tf = rng.uniform(0,1,(256,4)).astype(np.float32); tf[:,3]*=0.08; tf[:20,3]=0
vol = O.noise_u8(48,40,36,0x9E3779B9); ctx.load_volume(vol, tf)
for scale in ((1.57,1.0,1.0),(1,1,1)):
    for ert in (0, 2):
        opts = vv.make_options(ert_threshold=(2.0 if ert else 0.8), count_samples=True)
        got = ctx.render(160,90,cam("b",scale),options=opts); want,n = O.render(vol,tf,160,90,cam("b",scale),options=opts)
        report(f"colour scale{scale} ert{ert} n={n}/{ctx.last_sample_count()}", got, want)
