"""CPU checks of the measurement helpers that feed numbers into DESIGN.md / bench.py's JSON line (no GPU, no oracle): the opcode-mix pricing of
tools/valu_roofline.py on a hand-written ISA listing, and bench.py's rule that committed PMC-derived figures are printed only for the exact sources
they were measured on."""
import importlib.util
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

ISA = """
	.text
_ZN5vfhip6k_demoEv:                     ; @_ZN5vfhip6k_demoEv
; %bb.0:
	v_add_f32_e32 v0, v1, v2
	v_fma_f32 v0, v1, v2, v3
	v_mul_f32_e32 v0, s4, v2
	v_fmac_f32_e32 v0, 0x3f000000, v2
	v_perm_b32 v0, v1, v2, s5
	v_exp_f32_e32 v0, v1
	v_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[0:1]
	v_rndne_f32_e32 v0, v1
	s_mov_b32 s0, 1
	ds_read_b32 v1, v2
	s_endpgm
	.end_amdhsa_kernel
"""


def load_tool(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "tools", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    argv = sys.argv
    sys.argv = [name]                       # (the tools run main () at import: keep it from reading pytest's arguments; it fails harmlessly)
    try:
        spec.loader.exec_module(mod)
    except (SystemExit, IndexError, FileNotFoundError):
        pass
    finally:
        sys.argv = argv
    return mod


def test_valu_roofline_prices_the_three_classes(tmp_path):
    f = tmp_path / "demo.s"
    f.write_text(ISA)
    vr = load_tool("valu_roofline")
    mix = vr.kernel_mix(str(f))
    k = mix["_ZN5vfhip6k_demoEv"]
    # full rate: v_add, v_fmac with a literal; an SGPR operand on v_mul; v_fma (three sources) and v_perm half rate; one transcendental; one packed f32;
    # one rounding, which overlaps with scalar full-rate instructions when there are at least twice as many of them (2 >= 2 x 1 here)
    assert k["classes"] == {"full": 2, "full_sgpr": 1, "half": 2, "trans": 1, "pk_f32": 1, "side": 1}
    want = (2 * vr.C_FULL + vr.C_SGPR + 2 * vr.C_HALF + vr.C_TRANS + vr.C_PK + vr.C_SIDE_OVERLAPPED) / 8
    assert abs(k["avg_issue_cycles"] - want) < 1e-3 and k["valu_static"] == 8
    # ... and does not when they are fewer
    f2 = tmp_path / "demo2.s"
    f2.write_text(ISA.replace("\tv_add_f32_e32 v0, v1, v2\n", ""))
    k2 = vr.kernel_mix(str(f2))["_ZN5vfhip6k_demoEv"]
    assert abs(k2["avg_issue_cycles"] - (vr.C_FULL + vr.C_SGPR + 2 * vr.C_HALF + vr.C_TRANS + vr.C_PK + vr.C_SIDE) / 7) < 1e-3


def test_committed_pmc_figures_are_keyed_by_source(tmp_path, monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    assert len(bench.csrc_sha16()) == 16 and bench.csrc_sha16() == bench.csrc_sha16()
    prof = tmp_path / "profiles"
    prof.mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    (prof / "valu_latest.json").write_text(json.dumps({"source_sha16": "0" * 16, "kernels": {"k_x": {"valu_issue_share": 0.5, "avg_issue_cycles": 3.0}}}))
    assert bench.load_valu() == {}                                         # measured on other sources: not printed
    (prof / "valu_latest.json").write_text(json.dumps({"source_sha16": bench.csrc_sha16(), "kernels": {"k_x": {"valu_issue_share": 0.5, "avg_issue_cycles": 3.0}}}))
    assert bench.load_valu()["k_x"]["valu_issue_share"] == 0.5


def test_bench_configs_is_importable_without_a_gpu():
    sys.path.insert(0, ROOT)
    import bench_configs
    assert [n for n in ("c1", "c3", "c4", "c5", "others") if callable(getattr(bench_configs, n))] == ["c1", "c3", "c4", "c5", "others"]
