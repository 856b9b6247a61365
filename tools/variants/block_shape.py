# traffic_split.py --variants: block shape of march_kernel on C3 (32 x 2 wave tiles; waves stacked / 2 x 2 / side by side)
VARIANTS = [
    ("32x8 blocks (shipped)", {}, None),
    ("64x4 blocks", {"VV_BLOCK_W": "64"}, None),
    ("128x2 blocks", {"VV_BLOCK_W": "128"}, None),
    ("64x4, 3 blocks/CU", {"VV_BLOCK_W": "64", "VV_LDS_RESERVE": "49000"}, None),
    ("128x2, 3 blocks/CU", {"VV_BLOCK_W": "128", "VV_LDS_RESERVE": "49000"}, None),
    ("64x4, unroll 2", {"VV_BLOCK_W": "64", "VV_UNROLL": "2"}, None),
    ("64x4, xcd_band 2", {"VV_BLOCK_W": "64", "VV_XCD_BAND": "2"}, None),
    ("128x2, xcd_band 4", {"VV_BLOCK_W": "128", "VV_XCD_BAND": "4"}, None),
    ("32x8 again", {}, None),
]
