# traffic_split.py --view b --variants: block / strip shapes of the bricked kernel on the rotated camera (8 x 8 wave tiles)
VARIANTS = [
    ("base", {}, None),
    ("32x8 blocks", {"VV_BLOCK_W": "32"}, None),
    ("16x16 blocks", {"VV_BLOCK_W": "16"}, None),
    ("8x32 blocks", {"VV_BLOCK_W": "8"}, None),
    ("16x16, 3 blocks/CU", {"VV_BLOCK_W": "16", "VV_LDS_RESERVE": "49000"}, None),
    ("16x16, one round (rows 34-43)", {"VV_BLOCK_W": "16"}, (34, 43)),
    ("base again", {}, None),
]
