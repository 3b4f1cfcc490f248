#!/usr/bin/env python3
"""tools/isa_sgpr.py <file.s> [kernel-substring] — per kernel: VALU instructions by opcode, and how many of the FULL-RATE ones (add / sub / mul / fma / fmac /
logic / shifts / mov: 2.2 cycles per wave64 on VGPR or literal operands) carry an SGPR operand, which makes them issue in 3.9 (tools/ubench/valu_occ.hip)."""
import collections
import re
import sys

FULL = {"v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_fma_f32", "v_fmac_f32", "v_fmaak_f32", "v_fmamk_f32", "v_add_u32", "v_sub_u32", "v_subrev_u32",
        "v_and_b32", "v_or_b32", "v_xor_b32", "v_lshrrev_b32", "v_ashrrev_i32", "v_mov_b32", "v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32", "v_not_b32"}
s = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for mm in re.finditer(r"^(\S+):\s*; @\1\n(.*?)\.end_amdhsa_kernel", s, re.S | re.M):
    name, body = mm.group(1), mm.group(2)
    if pat not in name:
        continue
    tot, sc = collections.Counter(), collections.Counter()
    for line in body.splitlines():
        t = line.strip()
        if not t.startswith("v_"):
            continue
        op = t.split()[0].replace("_e32", "").replace("_e64", "")
        args = t[len(t.split()[0]):].split(";")[0]
        tot[op] += 1
        if re.search(r"(?<![\w.])s\d+\b|s\[\d+:\d+\]", args):
            sc[op] += 1
    full = sum(n for op, n in tot.items() if op in FULL)
    full_s = sum(n for op, n in sc.items() if op in FULL)
    print(f"{name}: VALU {sum(tot.values())}, full-rate {full}, of those with an SGPR operand {full_s}")
    for op, n in tot.most_common(12):
        print(f"    {op:22s} {n:5d}   sgpr operand {sc[op]:5d}")
