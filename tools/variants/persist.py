# (the kernel side of this experiment was removed again: see profiles/r03_traffic_split.txt section C and commit bca3ce6)
# traffic_split.py --variants: persistent neighbour-aligned tile order of march_kernel on C3
VARIANTS = [
    ("base (hardware dispatch, 2 blocks/CU)", {}, None),
    ("persist 2 blocks/CU", {"VV_PERSIST": "2"}, None),
    ("persist 3 blocks/CU", {"VV_PERSIST": "3", "VV_LDS_RESERVE": "49000"}, None),
    ("persist 4 blocks/CU", {"VV_PERSIST": "4", "VV_LDS_RESERVE": "36000"}, None),
    ("persist 2, unroll 2", {"VV_PERSIST": "2", "VV_UNROLL": "2"}, None),
    ("persist 3, unroll 2", {"VV_PERSIST": "3", "VV_LDS_RESERVE": "49000", "VV_UNROLL": "2"}, None),
    ("persist 1 block/CU", {"VV_PERSIST": "1", "VV_LDS_RESERVE": "155000"}, None),
    ("base again", {}, None),
]
