"""The CPU oracle (oracle/gst114.c) against golden vectors produced by the real GStreamer 1.14.0
`videoconvert ! videoscale` elements (tools/gen_goldens.py).  Bit-exact is the bar."""
import hashlib

import numpy as np
import pytest

import oracle_lib

MANIFEST, Z = oracle_lib.load_golden()

# sha256 recorded in SURVEY.md §8c for the two BASELINE configs (independent of this repo's generator run)
SURVEY_SHA = {
    "c1_vts_1080_to_640x480": ("4876147c45b75470aa60ec1b1a4a017218162e253898ca7744bd79b627933556",
                               "8b7dabdeb6ed05601e56a4c1a6040dcf73768452b68c64a8ccd6d6b747e1a17c"),
    "c2_vts_2160_to_1080": ("c2767da3b7c7bcca0e52290a7c6fccd876c3d45ea9a016aa9bd5cc2ceee28357",
                            "0ae25f64c20ebba8e7a44b36728672083e181c2bc5f5ad0e10d16475f448cf88"),
}


def test_fixture_integrity():
    assert len(MANIFEST) >= 60
    for c in MANIFEST:
        assert hashlib.sha256(Z[c["name"] + "_in"].tobytes()).hexdigest() == c["in_sha256"]
        assert hashlib.sha256(Z[c["name"] + "_out"].tobytes()).hexdigest() == c["out_sha256"]
    for name, (i, o) in SURVEY_SHA.items():
        c = next(c for c in MANIFEST if c["name"] == name)
        assert (c["in_sha256"], c["out_sha256"]) == (i, o)


@pytest.mark.parametrize("case", MANIFEST, ids=[c["name"] for c in MANIFEST])
def test_oracle_matches_gstreamer(oracle, case):
    c = case
    got = oracle.convertscale(c["in_format"], c["w"], c["h"], Z[c["name"] + "_in"], c["colorimetry"], c["chroma_site"],
                              c["method"], c["out_format"], c["ow"], c["oh"])
    want = Z[c["name"] + "_out"].reshape(c["oh"], c["ow"], 4)
    assert np.array_equal(got, want), f"max diff {np.abs(got.astype(int) - want.astype(int)).max()}"


def test_known_answers(oracle):
    """ORC matrix known answers (SURVEY.md §8c rule 3): Y=235,U=V=128 -> 253 (not 255); Y=128 -> 128."""
    import ctypes as C
    r, g, b = C.c_int(), C.c_int(), C.c_int()
    for m in range(3):
        oracle.lib.gst114_yuv_to_rgb(m, 235, 128, 128, C.byref(r), C.byref(g), C.byref(b))
        assert (r.value, g.value, b.value) == (253, 253, 253)
        oracle.lib.gst114_yuv_to_rgb(m, 128, 128, 128, C.byref(r), C.byref(g), C.byref(b))
        assert (r.value, g.value, b.value) == (128, 128, 128)
        oracle.lib.gst114_yuv_to_rgb(m, 16, 128, 128, C.byref(r), C.byref(g), C.byref(b))
        assert r.value == g.value == b.value


# ---- cells whose output is NV12 / I420 (golden vectors from the real elements, second fixture file) ----------------
MANIFEST_Y, ZY = oracle_lib.load_golden("convertscale_gst114_yuvout.npz")


def meaningful(fmt, w, h, raw):
    """the bytes GStreamer defines: rows without their stride padding"""
    out = []
    for i, (off, stride) in enumerate(oracle_lib.raw_layout(fmt, w, h)[0]):
        rows = h if i == 0 else (h + 1) // 2
        wb = w if i == 0 else (w + 1) // 2 * (2 if fmt == "NV12" else 1)
        out.append(np.asarray(raw[off: off + rows * stride]).reshape(rows, stride)[:, :wb].reshape(-1))
    return np.concatenate(out)


@pytest.mark.parametrize("case", MANIFEST_Y, ids=[c["name"] for c in MANIFEST_Y])
def test_oracle_matches_gstreamer_yuv_outputs(oracle, case):
    c = case
    got = oracle.convertscale(c["in_format"], c["w"], c["h"], ZY[c["name"] + "_in"], c["colorimetry"], c["chroma_site"],
                              c["method"], c["out_format"], c["ow"], c["oh"])
    want = ZY[c["name"] + "_out"]
    assert np.array_equal(meaningful(c["out_format"], c["ow"], c["oh"], got), meaningful(c["out_format"], c["ow"], c["oh"], want))
