# traffic_split.py --view b --variants: the bricked kernel on the rotated camera (theta 60, phi 36)
VARIANTS = [
    ("base (bricked copy, policy)", {}, None),
    ("1 block/CU", {"VV_LDS_RESERVE": "155000"}, None),
    ("2 blocks/CU", {"VV_LDS_RESERVE": "76000"}, None),
    ("3 blocks/CU", {"VV_LDS_RESERVE": "49000"}, None),
    ("4 blocks/CU", {"VV_LDS_RESERVE": "36000"}, None),
    ("6 blocks/CU", {"VV_LDS_RESERVE": "22000"}, None),
    ("unroll 3", {"VV_UNROLL": "3"}, None),
    ("tile 16x4", {"VV_TILE_LOG2W": "4", "VV_BRICKED": "1"}, None),
    ("tile 32x2", {"VV_TILE_LOG2W": "5", "VV_BRICKED": "1"}, None),
    ("xcd_band 2", {"VV_XCD_BAND": "2"}, None),
    ("xcd_band 0", {"VV_XCD_BAND": "0"}, None),
    ("rows 34-43 (one round)", {}, (34, 43)),
    ("linear layout", {"VV_BRICKED": "0"}, None),
]
