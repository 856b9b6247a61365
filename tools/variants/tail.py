# traffic_split.py [--view b] --variants: the batched post-ERT tail of march_kernel on / off
VARIANTS = [
    ("tail batch on (policy)", {}, None),
    ("tail batch off", {"VV_TAIL": "0"}, None),
    ("on again", {}, None),
    ("off again", {"VV_TAIL": "0"}, None),
]
