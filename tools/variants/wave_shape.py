# traffic_split.py --variants: wave tile shape at the shipped occupancy (2 blocks per CU, 3 samples per trip) on C3
B = {"VV_LDS_RESERVE": "76000", "VV_UNROLL": "3"}
def v(**kw):
    e = dict(B); e.update({k: str(x) for k, x in kw.items()}); return e
VARIANTS = [
    ("policy: 32x2 waves, 64x4 blocks", {}, None),
    ("16x4 waves, 32x8 blocks", v(VV_TILE_LOG2W=4, VV_BLOCK_W=32), None),
    ("16x4 waves, 64x4 blocks", v(VV_TILE_LOG2W=4, VV_BLOCK_W=64), None),
    ("8x8 waves, 32x8 blocks (linear layout)", v(VV_TILE_LOG2W=3, VV_BRICKED=0), None),
    ("32x2 waves, 32x8 blocks", v(VV_BLOCK_W=32), None),
    ("32x2 waves, 64x4 blocks, 2 samples/trip", {"VV_UNROLL": "2"}, None),
    ("policy again", {}, None),
]
