#!/usr/bin/env python3
"""Table of a bench line's byte counts per march record: algorithmic (8^3 bricks), compulsory lines, lines of every issued gather, per-block footprints,
traffic at the L2's memory side.   usage: python3 tools/line_footprints_report.py profiles/r05_bench_line.json"""
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
rows = [("view a (headline)", j["roofline"], j.get("kernel_ms_rank0"))]
for k, name in (("side_view", "side view (z-fastest copy)"), ("rotated_view", "rotated view (bricked copy)"), ("phong", "view a + Phong")):
    if k in j and "roofline" in j[k]:
        rows.append((name, j[k]["roofline"], j[k].get("ms_per_frame")))
print("# C3 (1024^3 f32, 1920x1080, step 1/512): bytes per frame in GB; x = relative to `compulsory lines`")
print(f"{'record':30s} {'ms':>7s} {'algorithmic':>12s} {'compulsory':>11s} {'issued':>8s} {'per block':>10s} {'traffic':>8s}   issued x  block x  traffic x  traffic / block")
for name, r, ms in rows:
    a, c, i, b, t = (r.get(k) for k in ("algorithmic_bytes_per_launch", "compulsory_line_bytes", "issued_line_bytes", "block_line_bytes", "traffic"))
    g = lambda v: f"{v / 1e9:.3f}" if v else "-"
    x = lambda v: f"{v / c:.3f}" if (v and c) else "-"
    print(f"{name:30s} {ms:7.4f} {g(a):>12s} {g(c):>11s} {g(i):>8s} {g(b):>10s} {g(t):>8s}   {x(i):>8s} {x(b):>8s} {x(t):>10s}  {(t / b if t and b else 0):.3f}")
