# traffic_split.py --variants: the slab sweep (per-wave skewed lock-step consumer) against the gather kernel on C3
VARIANTS = [
    ("base (gather kernel)", {}, None),
    ("sweep default", {"VV_SWEEP": "1"}, None),
    ("sweep 64x8", {"VV_SWEEP": "1", "VV_SWEEP_WX": "2", "VV_SWEEP_WY": "4"}, None),
    ("sweep 64x6", {"VV_SWEEP": "1", "VV_SWEEP_WX": "2", "VV_SWEEP_WY": "3"}, None),
    ("sweep 64x4", {"VV_SWEEP": "1", "VV_SWEEP_WX": "2", "VV_SWEEP_WY": "2"}, None),
    ("sweep 96x4", {"VV_SWEEP": "1", "VV_SWEEP_WX": "3", "VV_SWEEP_WY": "2"}, None),
    ("sweep 128x4", {"VV_SWEEP": "1", "VV_SWEEP_WX": "4", "VV_SWEEP_WY": "2"}, None),
    ("sweep 32x8", {"VV_SWEEP": "1", "VV_SWEEP_WX": "1", "VV_SWEEP_WY": "4"}, None),
    ("sweep 64x6 nl=2", {"VV_SWEEP": "1", "VV_SWEEP_WX": "2", "VV_SWEEP_WY": "3", "VV_SWEEP_NL": "2"}, None),
    ("sweep 64x6 nl=4", {"VV_SWEEP": "1", "VV_SWEEP_WX": "2", "VV_SWEEP_WY": "3", "VV_SWEEP_NL": "4"}, None),
    ("sweep 64x6 group=2", {"VV_SWEEP": "1", "VV_SWEEP_WX": "2", "VV_SWEEP_WY": "3", "VV_SWEEP_GROUP": "2"}, None),
]
