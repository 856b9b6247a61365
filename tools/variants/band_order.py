# traffic_split.py --variants: dispatch order of the tiles of an XCD band (64 x 4 blocks by policy)
VARIANTS = [
    ("64x4, band 1 (policy)", {}, None),
    ("band 2, strip by strip", {"VV_XCD_BAND": "2"}, None),
    ("band 2, column by column", {"VV_XCD_BAND": "2", "VV_BAND_COLMAJOR": "1"}, None),
    ("band 4, column by column", {"VV_XCD_BAND": "4", "VV_BAND_COLMAJOR": "1"}, None),
    ("band 8, column by column", {"VV_XCD_BAND": "8", "VV_BAND_COLMAJOR": "1"}, None),
    ("32x8, band 2, column by column", {"VV_BLOCK_W": "32", "VV_XCD_BAND": "2", "VV_BAND_COLMAJOR": "1"}, None),
    ("32x8, band 4, column by column", {"VV_BLOCK_W": "32", "VV_XCD_BAND": "4", "VV_BAND_COLMAJOR": "1"}, None),
    ("128x2, band 4, column by column", {"VV_BLOCK_W": "128", "VV_XCD_BAND": "4", "VV_BAND_COLMAJOR": "1"}, None),
    ("128x2, band 8, column by column", {"VV_BLOCK_W": "128", "VV_XCD_BAND": "8", "VV_BAND_COLMAJOR": "1"}, None),
    ("policy again", {}, None),
]
