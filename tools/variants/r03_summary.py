# the kernels of round 3 side by side on C3 (traffic_split.sh): shipped gather kernel, skewed lock step, slab sweep v3
S = {"VV_SWEEP": "1"}
def sw(wx, wy, u, a):
    e = dict(S); e.update({"VV_SWEEP_WX": str(wx), "VV_SWEEP_WY": str(wy), "VV_SWEEP_STEPS": str(u), "VV_SWEEP_AHEAD": str(a)})
    return e
VARIANTS = [
    ("march_kernel (shipped)", {}, None),
    ("march_skew_kernel U=3", {"VV_SKEW": "3"}, None),
    ("sweep_kernel 64x8 U2 ahead 2", sw(2, 4, 2, 2), None),
    ("sweep_kernel 96x6 U2 ahead 2", sw(3, 3, 2, 2), None),
    ("sweep_kernel 64x8 U1 ahead 4", sw(2, 4, 1, 4), None),
]
