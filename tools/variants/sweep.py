# traffic_split.py --variants: the slab sweep (barrier-synchronous, block-wide skewed lock step) against the gather kernel on C3
S = {"VV_SWEEP": "1"}
def sw(wx, wy, **kw):
    e = dict(S); e["VV_SWEEP_WX"] = str(wx); e["VV_SWEEP_WY"] = str(wy)
    e.update({k: str(v) for k, v in kw.items()})
    return e
VARIANTS = [
    ("base (gather kernel)", {}, None),
    ("sweep default", dict(S), None),
    ("sweep 64x8 ahead 1", sw(2, 4, VV_SWEEP_AHEAD=1), None),
    ("sweep 64x8 ahead 3", sw(2, 4, VV_SWEEP_AHEAD=3), None),
    ("sweep 64x6", sw(2, 3), None),
    ("sweep 64x4", sw(2, 2), None),
    ("sweep 96x4", sw(3, 2), None),
    ("sweep 96x6", sw(3, 3), None),
    ("sweep 128x4", sw(4, 2), None),
    ("sweep 32x8", sw(1, 4), None),
    ("sweep 64x12", sw(2, 6), None),
    ("sweep 32x16", sw(1, 8), None),
]
