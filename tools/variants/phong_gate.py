# traffic_split.py --phong --variants: per-CU refresh gate of march_phong_kernel (C3 + Phong)
VARIANTS = [("phong base (3 blocks/CU)", {}, None)]
for res, nb in ((60000, 2), (30000, 3), (22000, 4), (13000, 5)):
    for g in (0, 1, 2, 3):
        if g >= nb and g: continue
        e = {"VV_LDS_RESERVE_PHONG": str(res)}
        if g: e["VV_PHONG_GATE"] = str(g)
        VARIANTS.append((f"phong {nb} blocks/CU gate {g}", e, None))
