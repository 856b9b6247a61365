# traffic_split.py --variants: skewed lock step (march_skew_kernel) against the lock-step kernel on C3
VARIANTS = [
    ("base", {}, None),
    ("skew U=3", {"VV_SKEW": "3"}, None),
    ("skew U=2", {"VV_SKEW": "3", "VV_UNROLL": "2"}, None),
    ("skew U=1", {"VV_SKEW": "3", "VV_UNROLL": "1"}, None),
    ("skew U=3, 3 blocks/CU", {"VV_SKEW": "3", "VV_LDS_RESERVE": "49000"}, None),
    ("skew U=2, 3 blocks/CU", {"VV_SKEW": "3", "VV_UNROLL": "2", "VV_LDS_RESERVE": "49000"}, None),
    ("skew U=2, 4 blocks/CU", {"VV_SKEW": "3", "VV_UNROLL": "2", "VV_LDS_RESERVE": "36000"}, None),
    ("skew U=1, 4 blocks/CU", {"VV_SKEW": "3", "VV_UNROLL": "1", "VV_LDS_RESERVE": "36000"}, None),
    ("skew U=1, 6 blocks/CU", {"VV_SKEW": "3", "VV_UNROLL": "1", "VV_LDS_RESERVE": "22000"}, None),
    ("skew U=3, rows 34-43 (one round)", {"VV_SKEW": "3"}, (34, 43)),
    ("base again", {}, None),
]
