#!/usr/bin/env python3
"""Rough VALU issue-cost model of one kernel's ISA (costs from scripts/ubench_valu.hip, cycles @2.4 GHz wall)."""
import re, sys, collections
f, kern = sys.argv[1], sys.argv[2]
txt = open(f).read()
m = re.search(r"^(_Z[A-Za-z0-9_]*%s[A-Za-z0-9_]*):[^\n]*\n(.*?)s_endpgm" % kern, txt, re.S | re.M)
body = m.group(2)
cost = collections.Counter(); cnt = collections.Counter()
for line in body.splitlines():
    t = line.strip().split()
    if not t or not t[0].startswith("v_"): continue
    op = t[0]
    if "dpp" in op or "wave_sh" in line or "row_sh" in line: k, c = "dpp", 4.4
    elif op.startswith(("v_rcp", "v_exp", "v_log", "v_sqrt", "v_rsq", "v_sin", "v_cos")): k, c = "trans", 8.2
    elif op.startswith(("v_div_", "v_med3", "v_lshl_add_u64", "v_mad_", "v_mul_lo", "v_mul_hi", "v_add3", "v_lshlrev_b64", "v_or3", "v_and_or", "v_bfe", "v_min3", "v_max3")): k, c = "vop3x", 4.4
    elif op.startswith(("v_fma", "v_fmac")): k, c = "fma", 3.0
    elif op.endswith("_e64"): k, c = "e64", 3.8
    elif op.startswith("v_cmp"): k, c = "cmp", 3.0
    else: k, c = "plain", 2.5
    cnt[k] += 1; cost[k] += c
tot = sum(cost.values())
print(f"{m.group(1)[:70]}: VALU instrs {sum(cnt.values())}, model cycles {tot:.0f}")
for k in sorted(cnt, key=lambda k: -cost[k]):
    print(f"   {k:6s} n={cnt[k]:5d} cycles={cost[k]:8.0f} ({100*cost[k]/tot:4.1f}%)")
