# march_phong_pair_kernel against march_phong_kernel on C3 + Phong (tools/traffic_split.sh <tag> --phong --variants tools/variants/phong_pair.py)
VARIANTS = [
    ("march_phong_kernel, 3 blocks/CU (shipped since round 2)", {"VV_PHONG_PAIR": "0"}, None),
    ("march_phong_kernel, 2 blocks/CU", {"VV_PHONG_PAIR": "0", "VV_LDS_RESERVE_PHONG": "55000"}, None),
    ("pair kernel, 3 blocks/CU (6 slabs)", {"VV_PHONG_PAIR": "1", "VV_LDS_RESERVE_PHONG": "30000"}, None),
    ("pair kernel, 2 blocks/CU (4 slabs)", {"VV_PHONG_PAIR": "1", "VV_LDS_RESERVE_PHONG": "55000"}, None),
    ("pair kernel, 1 block/CU (2 slabs)", {"VV_PHONG_PAIR": "1", "VV_LDS_RESERVE_PHONG": "100000"}, None),
]
