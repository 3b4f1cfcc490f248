#!/usr/bin/env python3
"""tools/isa_hist.py <file.s> <kernel-substring> — opcode histogram + wave64 VALU issue cycles of a kernel's hottest
basic block (the loop body), using the per-opcode issue costs measured on gfx950 (profiles/r01b_valu_rate2.txt,
gpurun_out r02a: 2 cycles for add/sub/logic/shift-right/mov/f32 add-mul-fma and the 16-bit VOP2 forms, 8 for
v_ashr_pk_u8_i32 / v_mad_u16 / v_fma_f16, 4 for every other VALU op)."""
import collections
import re
import sys

TWO = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_lshrrev_b32", "v_ashrrev_i32", "v_mov_b32",
       "v_add_f32", "v_sub_f32", "v_mul_f32", "v_fma_f32", "v_fmac_f32", "v_add_u16", "v_sub_u16", "v_mul_lo_u16", "v_max_u16", "v_lshlrev_b16",
       "v_ashrrev_i16", "v_mul_f16", "v_add_f16", "v_not_b32", "v_subrev_f32"}
EIGHT = {"v_ashr_pk_u8_i32", "v_mad_u16", "v_fma_f16"}


def cost(op):
    op = op.replace("_e32", "").replace("_e64", "")
    if not op.startswith("v_"):
        return 0
    return 2 if op in TWO else 8 if op in EIGHT else 4


def main():
    s = open(sys.argv[1]).read()
    pat = sys.argv[2]
    m = None
    for mm in re.finditer(r"^(\S+):\s*; @\1\n(.*?)\.end_amdhsa_kernel", s, re.S | re.M):
        if pat in mm.group(1):
            m = mm
            break
    if not m:
        sys.exit("kernel not found")
    body = m.group(2)
    blocks = re.split(r"^\.LBB\d+_\d+:.*$", body, flags=re.M)
    want = int(sys.argv[3]) if len(sys.argv) > 3 else None
    order = sorted(range(len(blocks)), key=lambda i: -len(blocks[i]))
    big = blocks[order[0] if want is None else want]
    ops = collections.Counter()
    for line in big.splitlines():
        line = line.strip()
        if not line or line[0] in ";.":
            continue
        ops[line.split()[0]] += 1
    cyc = sum(cost(k) * v for k, v in ops.items())
    print(m.group(1), "block sizes (instr):", [sum(1 for l in blocks[i].splitlines() if l.strip() and l.strip()[0] not in ";.") for i in order[:6]])
    for k, v in ops.most_common():
        print(f"  {k:30s} {v:4d} x{cost(k)}")
    print("  instructions:", sum(ops.values()), " VALU:", sum(v for k, v in ops.items() if k.startswith("v_")), " VALU issue cycles (wave64):", cyc)
    for l in body.splitlines():
        if re.search(r"; (NumVgprs|NumSgprs|Occupancy|ScratchSize|LDSByteSize)", l):
            print(" ", l.strip())


main()
