VARIANTS = [
    ("base", {}, None),
    ("skew U=3", {"VV_SKEW": "3"}, None),
    ("skew U=2", {"VV_SKEW": "3", "VV_UNROLL": "2"}, None),
    ("skew U=3, 3 blocks/CU", {"VV_SKEW": "3", "VV_LDS_RESERVE": "49000"}, None),
    ("skew U=2, 3 blocks/CU", {"VV_SKEW": "3", "VV_UNROLL": "2", "VV_LDS_RESERVE": "49000"}, None),
    ("skew U=3, rows 34-43 (one round)", {"VV_SKEW": "3"}, (34, 43)),
    ("base, rows 34-43 (one round)", {}, (34, 43)),
]
