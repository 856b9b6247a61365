# traffic_split.py --variants: the slab sweep (barrier-synchronous, block-wide skewed lock step) against the gather kernel on C3
S = {"VV_SWEEP": "1"}
def sw(wx, wy, u, a):
    e = dict(S); e.update({"VV_SWEEP_WX": str(wx), "VV_SWEEP_WY": str(wy), "VV_SWEEP_STEPS": str(u), "VV_SWEEP_AHEAD": str(a)})
    return e
VARIANTS = [("base (gather kernel)", {}, None), ("sweep default", dict(S), None)]
for wx, wy in ((2, 4), (2, 3), (2, 2), (3, 2), (3, 3), (1, 4)):
    for u, a in ((1, 4), (1, 8), (2, 2), (2, 4)):
        VARIANTS.append((f"sweep {32 * wx}x{2 * wy} U{u} ahead {a}", sw(wx, wy, u, a), None))
