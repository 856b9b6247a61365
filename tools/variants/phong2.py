# march_phong2_kernel against march_phong_kernel on C3 + Phong (tools/traffic_split.sh r4_phong2 --phong --variants tools/variants/phong2.py)
VARIANTS = [
    ("v1, 3 blocks/CU (round 3's kernel and policy)", {"VV_PHONG2": "0"}, None),
    ("v2 S=1, 3 blocks/CU", {"VV_PHONG2": "1"}, None),
    ("v2 S=1, 2 blocks/CU", {"VV_PHONG2": "1", "VV_LDS_RESERVE_PHONG": "55000"}, None),
    ("v2 S=2, 2 blocks/CU", {"VV_PHONG2": "2", "VV_LDS_RESERVE_PHONG": "30000"}, None),
    ("v2 S=2, 1 block/CU", {"VV_PHONG2": "2", "VV_LDS_RESERVE_PHONG": "60000"}, None),
]
