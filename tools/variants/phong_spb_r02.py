# (run against round 2's library: tools/run_with_lib.py) -- what did two slabs per Phong block do to the BYTES?
VARIANTS = [
    ("r02 phong, 1 slab per block", {"VV_PHONG_SPB": "1"}, None),
    ("r02 phong, 2 slabs per block", {"VV_PHONG_SPB": "2"}, None),
    ("r02 phong, 2 slabs, 2 blocks/CU", {"VV_PHONG_SPB": "2", "VV_LDS_RESERVE_PHONG": "50000"}, None),
]
