import json, sys
for l in open(sys.argv[1]):
    if not l.startswith("{"): continue
    j = json.loads(l); c = j["counters"]
    print(f"{j['name']:32s} {j['ms']:7.3f} ms samples {j['samples']} miss {c[4]} staged {c[5] / 1e9:.2f} GB trips {c[6]} gather-trips {c[13]} bails {c[7] >> 48} err {c[7] & 0xffffffffffff:#x}")
