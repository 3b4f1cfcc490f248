#!/usr/bin/env python3
"""tools/valu_roofline.py <pmc dir with pmc_summary.json> <file.s> [more .s ...] — a VALU roofline next to the HBM one for kernels the vector ALU bounds.

  valu_cycles  = SQ_INSTS_VALU (wave-instructions per dispatch, PMC) x the kernel's average ISSUE cycles per wave64 instruction
  issue_share  = valu_cycles / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs)          (cannot exceed 1 when the class costs are right)
  valu_floor   = the dispatch time if the SIMDs did nothing but issue these instructions = measured time x issue_share

The average issue cost comes from the kernel's opcode mix in the ISA listing (static: the hot loops dominate; uniform-branch alternatives the
workload does not take are in it too, which the README says where it matters), with the classes MEASURED on gfx950 (tools/ubench/valu_occ.hip,
valu_rate3.hip, valu_mix.hip; profiles/r03j_valu_occ.txt, r03a_valu_rate3.txt, r03ah_valu_mix.txt), in cycles per wave64 instruction with the
chip's own v_add_u32 = 2.2:
  full rate   2.0 : add / sub / mul / fmac / fmaak / fmamk / logic / shift-right / mov on VGPR, inline-constant or literal operands
  SGPR        3.9 : a full-rate opcode with an SGPR operand
  side unit   3.4 : conversions and roundings (v_cvt_f32_ubyte*, v_cvt_pk_u8_f32, v_rndne / floor / fract / trunc ...) — but 2.0 in a kernel that has at
                    least twice as many scalar full-rate instructions: they execute BESIDE those (v_cvt_f32_ubyte + v_fmac alternating: 3.9 per pair)
  half rate   3.6 : every other VALU opcode (three-source and byte opcodes: v_fma_f32, v_mad_i32_i16, v_perm_b32, v_lerp_u8, v_dot4, SDWA forms,
                    v_cndmask, min / max ...); nothing overlaps with these
  transcend.  6.75: v_exp / v_log / v_rcp / v_rsq / v_sqrt / v_sin / v_cos            packed f32 (v_pk_*_f32): 3.6 per instruction (two lanes' worth)
Round 3 first priced full rate at 2.2 and everything else at 3.9: the restructured deinterlacer (scalar f32 with 20 % conversions / roundings) then read
1.11 — the overlap above is what that model lacked.
SQ_ACTIVE_INST_VALU is NOT used: it counts quad-cycles, so every full-rate instruction reads as 4 cycles and `x 4 / SIMD-cycles` came out at 1.1-1.4
in round 2: that figure is an over-count, not a utilisation."""
import collections
import json
import os
import re
import sys

FULL = {"v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_fmac_f32", "v_fmaak_f32", "v_fmamk_f32", "v_add_u32", "v_sub_u32", "v_subrev_u32",
        "v_and_b32", "v_or_b32", "v_xor_b32", "v_lshrrev_b32", "v_ashrrev_i32", "v_mov_b32", "v_not_b32", "v_add_u16", "v_sub_u16", "v_mul_lo_u16", "v_max_u16",
        "v_lshlrev_b16", "v_ashrrev_i16", "v_mul_f16", "v_add_f16", "v_mul_legacy_f32"}
TRANS = {"v_exp_f32", "v_log_f32", "v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_sin_f32", "v_cos_f32", "v_rcp_iflag_f32"}
SIDE = {"v_cvt_f32_ubyte0", "v_cvt_f32_ubyte1", "v_cvt_f32_ubyte2", "v_cvt_f32_ubyte3", "v_cvt_pk_u8_f32", "v_rndne_f32", "v_floor_f32", "v_fract_f32", "v_trunc_f32",
        "v_ceil_f32", "v_cvt_f32_u32", "v_cvt_f32_i32", "v_cvt_u32_f32", "v_cvt_i32_f32", "v_cvt_flr_i32_f32", "v_cvt_rpi_i32_f32", "v_cvt_f32_f16", "v_cvt_f16_f32"}
C_FULL, C_SGPR, C_SIDE, C_SIDE_OVERLAPPED, C_HALF, C_TRANS, C_PK = 2.0, 3.9, 3.4, 2.0, 3.6, 6.75, 3.6


def kernel_mix(path):
    out = {}
    s = open(path).read()
    for mm in re.finditer(r"^(\S+):\s*; @\1\n(.*?)\.end_amdhsa_kernel", s, re.S | re.M):
        cls = collections.Counter()
        for line in mm.group(2).splitlines():
            t = line.strip()
            if not t.startswith("v_"):
                continue
            op = t.split()[0].replace("_e32", "").replace("_e64", "").replace("_sdwa", "#sdwa").replace("_dpp", "#dpp")
            args = t[len(t.split()[0]):].split(";")[0]
            sgpr = re.search(r"(?<![\w.])s\d+\b|s\[\d+:\d+\]", args) is not None
            if op in TRANS:
                cls["trans"] += 1
            elif op.startswith("v_pk_") and op.endswith("_f32"):
                cls["pk_f32"] += 1
            elif op in FULL and not sgpr:
                cls["full"] += 1
            elif op in FULL:
                cls["full_sgpr"] += 1
            elif op in SIDE:
                cls["side"] += 1
            else:
                cls["half"] += 1
        n = sum(cls.values())
        if n:
            c_side = C_SIDE_OVERLAPPED if cls["full"] >= 2 * cls["side"] else C_SIDE
            avg = (cls["full"] * C_FULL + cls["full_sgpr"] * C_SGPR + cls["side"] * c_side + cls["half"] * C_HALF + cls["trans"] * C_TRANS + cls["pk_f32"] * C_PK) / n
            out[mm.group(1)] = {"valu_static": n, "classes": dict(cls), "avg_issue_cycles": round(avg, 3)}
    return out


def demangle_key(name):
    m = re.match(r"_ZN5vfhip\d+(k_[a-z0-9_]+?)(I|E|ENS)", name)
    return m.group(1) if m else name


def main():
    pmc = json.load(open(os.path.join(sys.argv[1], "pmc_summary.json")))
    mixes = {}
    for f in sys.argv[2:]:
        mixes.update(kernel_mix(f))
    res = {}
    for kname, c in pmc.items():
        g = {n: v["avg"] for n, v in c.items()}
        if "SQ_INSTS_VALU" not in g or "GRBM_GUI_ACTIVE" not in g:
            continue
        short = kname.split("::")[-1].split("<")[0]
        # template arguments as they appear in the PMC name: <true, false> ...; pick the mangled instantiation whose bools match when there are several
        cands = [(m, v) for m, v in mixes.items() if demangle_key(m) == short]
        if not cands:
            continue
        targs = re.findall(r"true|false|-?\d+", kname.split("<", 1)[1]) if "<" in kname else []
        def score(m):
            enc = "".join("Lb1E" if a == "true" else "Lb0E" if a == "false" else f"Li{a}E" for a in targs)
            return 0 if enc and enc in m else 1
        mname, mix = sorted(cands, key=lambda mv: score(mv[0]))[0]
        simd_cycles = g["GRBM_GUI_ACTIVE"] / 8 * 1024
        valu_cycles = g["SQ_INSTS_VALU"] * mix["avg_issue_cycles"]
        res[kname] = {"isa": mname, "valu_wave_instructions_per_dispatch": round(g["SQ_INSTS_VALU"]), "avg_issue_cycles": mix["avg_issue_cycles"], "classes": mix["classes"],
                      "kernel_cycles": round(g["GRBM_GUI_ACTIVE"] / 8), "valu_issue_share": round(valu_cycles / simd_cycles, 3),
                      "note": "valu_floor = measured kernel time x valu_issue_share"}
        if "SQ_ACTIVE_INST_VALU" in g:
            res[kname]["sq_active_inst_valu_x4_over_simd_cycles (over-count, for the record)"] = round(g["SQ_ACTIVE_INST_VALU"] * 4 / simd_cycles, 3)
    print(json.dumps(res, indent=1))


main()
