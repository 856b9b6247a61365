# march_phong2_kernel against march_phong_kernel on C3 + Phong (tools/traffic_split.sh <tag> --phong --variants tools/variants/phong2.py)
VARIANTS = [
    ("v1, 3 blocks/CU (round 3's kernel and policy)", {"VV_PHONG2": "0"}, None),
    ("v3 S=1, 3 blocks/CU", {"VV_PHONG2": "1"}, None),
    ("v3 S=1, 4 blocks/CU", {"VV_PHONG2": "1", "VV_LDS_RESERVE_PHONG": "20000"}, None),
    ("v3 S=1, 2 blocks/CU", {"VV_PHONG2": "1", "VV_LDS_RESERVE_PHONG": "55000"}, None),
    ("v3 S=2, 2 blocks/CU (4 slabs)", {"VV_PHONG2": "2", "VV_LDS_RESERVE_PHONG": "30000"}, None),
    ("v3 S=2, 1 block/CU (2 slabs)", {"VV_PHONG2": "2", "VV_LDS_RESERVE_PHONG": "70000"}, None),
]
