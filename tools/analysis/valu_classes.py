#!/usr/bin/env python3
"""Developer aid: classify the vector instructions of an ISA listing (hipcc -S) by issue class as measured on MI355X (tools/ubench/int_rate.hip):
fast (add / sub / mul / fma / fmac f32, add / sub u32, and / or / xor, mov: ~1.0), the same with an SGPR operand (1.65), everything else (1.65-1.8),
transcendentals (3.3).  usage: valu_classes.py file.s [first_line last_line]"""
import re, sys
FAST = {"v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_fma_f32", "v_fmac_f32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_mov_b32"}
TRANS = {"v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_exp_f32", "v_log_f32", "v_sin_f32", "v_cos_f32", "v_rcp_iflag_f32"}
lines = open(sys.argv[1]).read().split("\n")
lo, hi = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1, len(lines))
n = {"fast": 0, "fast+sgpr": 0, "slow": 0, "trans": 0}
ops = {}
for l in lines[lo - 1:hi]:
    m = re.match(r"\s+(v_[a-z0-9_]+?)(_e32|_e64|_dpp|_sdwa)?\s+(.*?)(;.*)?$", l)
    if not m:
        continue
    op, args = m.group(1), m.group(3)
    srcs = args.split(",")[1:]
    sg = any(re.match(r"\s*(s\[?\d|vcc|exec|m0)", a) for a in srcs)
    if op in TRANS: k = "trans"
    elif op in FAST: k = "fast+sgpr" if sg else "fast"
    else: k = "slow"
    n[k] += 1
    if k != "fast": ops[(k, op)] = ops.get((k, op), 0) + 1
cost = n["fast"] + 1.65 * (n["fast+sgpr"] + n["slow"]) + 3.3 * n["trans"]
print(n, "total", sum(n.values()), "cost in fast-instruction units %.1f" % cost)
for (k, op), c in sorted(ops.items(), key=lambda t: -t[1])[:25]:
    print(f"  {k:10s} {op:20s} {c}")
