"""GPU parity tests of vfhipconvertscale through the C ABI (libvfhip.so), gst-exact numerics.

  * every golden vector made by the real GStreamer 1.14 elements (tests/golden, bit-exact);
  * seeded random frames against the CPU oracle (oracle/gst114.c), incl. ragged / tiny / odd sizes;
  * BASELINE sizes (2160p -> 1080p batches) through size-independent properties.
Bar: bit-exact (integer path)."""
import numpy as np
import pytest

import oracle_lib

pytestmark = pytest.mark.gpu
MANIFEST, Z = oracle_lib.load_golden()


def run(vfhip, c_in_fmt, w, h, raw, col, site, method, ofmt, ow, oh, numerics="gst-exact"):
    if col is None:
        col, site = oracle_lib.default_colorimetry(h)
    cs = vfhip.ConvertScale(0)
    cs.configure(c_in_fmt, w, h, ofmt, ow, oh, method=method, colorimetry=col, chroma_site=site, numerics=numerics)
    out = cs.process(raw)
    name = cs.kernel_name
    cs.close()
    return (out.reshape(oh, ow, 4) if ofmt in ("BGRA", "RGBA") else out), name


@pytest.mark.parametrize("case", MANIFEST, ids=[c["name"] for c in MANIFEST])
def test_golden_gstreamer_vectors(vfhip, case):
    c = case
    got, kname = run(vfhip, c["in_format"], c["w"], c["h"], Z[c["name"] + "_in"], c["colorimetry"], c["chroma_site"],
                     c["method"], c["out_format"], c["ow"], c["oh"])
    want = Z[c["name"] + "_out"].reshape(c["oh"], c["ow"], 4)
    assert np.array_equal(got, want), f"{kname}: max diff {np.abs(got.astype(int) - want.astype(int)).max()}"
    if c["name"] == "c2_vts_2160_to_1080":
        assert kname == "k_cs_nv12_half"          # the headline config must take the fast path


HALF_CASES = [(64, 36), (8, 6), (16, 8), (1024, 40), (200, 72), (3840, 64), (520, 34)]


@pytest.mark.parametrize("w,h", HALF_CASES)
@pytest.mark.parametrize("col,site", [("bt601", "jpeg"), ("bt709", "mpeg2"), ("bt2020", "mpeg2"), ("bt2020", "jpeg")])
@pytest.mark.parametrize("ofmt", ["BGRA", "RGBA"])
def test_half_kernel_vs_oracle(vfhip, oracle, w, h, col, site, ofmt):
    rng = np.random.default_rng(w * 1000 + h)
    _, size = vfhip.plane_layout("NV12", w, h)
    raw = rng.integers(0, 256, size, dtype=np.uint8)
    got, kname = run(vfhip, "NV12", w, h, raw, col, site, "bilinear", ofmt, w // 2, h // 2)
    assert kname == "k_cs_nv12_half"
    want = oracle.convertscale("NV12", w, h, raw, col, site, "bilinear", ofmt, w // 2, h // 2)
    assert np.array_equal(got, want), f"max diff {np.abs(got.astype(int) - want.astype(int)).max()}"


def test_extreme_values_half_kernel(vfhip, oracle):
    """saturation paths of the ORC matrix: all-0, all-255 and alternating extremes"""
    w, h = 64, 16
    _, size = vfhip.plane_layout("NV12", w, h)
    for fill in (0, 255, None):
        raw = np.full(size, fill, np.uint8) if fill is not None else np.tile(np.array([0, 255, 255, 0, 16, 235, 240, 1], np.uint8), size // 8 + 1)[:size]
        for col in ("bt601", "bt709", "bt2020"):
            got, _ = run(vfhip, "NV12", w, h, raw, col, "mpeg2", "bilinear", "BGRA", w // 2, h // 2)
            want = oracle.convertscale("NV12", w, h, raw, col, "mpeg2", "bilinear", "BGRA", w // 2, h // 2)
            assert np.array_equal(got, want)


def test_yuv_cube_sweep(vfhip, oracle):
    """every (Y, U, V) on a 16-step lattice + the limits, same-size conversion (generic kernel)"""
    vals = np.array(sorted(set(list(range(0, 256, 15)) + [16, 128, 235, 240, 255])), np.uint8)
    n = len(vals)
    w, h = 2 * n * n, 2 * n          # chroma sample (j,k): U=vals[k%n], V=vals[k//n]; luma row pair j: Y=vals[j]
    ys, hp = vfhip.r4(w), h
    raw = np.zeros(ys * hp + ys * (hp // 2), np.uint8)
    y = raw[:ys * h].reshape(h, ys)
    uv = raw[ys * hp:].reshape(hp // 2, ys)
    for j in range(n):
        y[2 * j:2 * j + 2, :w] = vals[j]
        uv[j, 0:w:2] = np.tile(vals, n)
        uv[j, 1:w:2] = np.repeat(vals, n)
    for col in ("bt601", "bt709", "bt2020"):
        for fmt in ("NV12",):
            got, _ = run(vfhip, fmt, w, h, raw, col, "jpeg", "bilinear", "BGRA", w, h)
            want = oracle.convertscale(fmt, w, h, raw, col, "jpeg", "bilinear", "BGRA", w, h)
            assert np.array_equal(got, want)


@pytest.mark.parametrize("seed", range(12))
def test_random_sizes_vs_oracle(vfhip, oracle, seed):
    rng = np.random.default_rng(1000 + seed)
    fmt = ["NV12", "I420"][seed % 2]
    w, h, ow, oh = (int(v) for v in rng.integers(1, 200, 4))
    w, h = max(w, 2), max(h, 2)
    col = ["bt601", "bt709", "bt2020"][seed % 3]
    site = ["jpeg", "mpeg2"][(seed // 3) % 2]
    method = "nearest" if seed % 5 == 4 else "bilinear"
    _, size = vfhip.plane_layout(fmt, w, h)
    raw = rng.integers(0, 256, size, dtype=np.uint8)
    got, _ = run(vfhip, fmt, w, h, raw, col, site, method, "BGRA", ow, oh)
    want = oracle.convertscale(fmt, w, h, raw, col, site, method, "BGRA", ow, oh)
    assert np.array_equal(got, want), (fmt, w, h, ow, oh, col, site, method)


def test_rgb_to_rgb_scale_and_swizzle(vfhip, oracle):
    import ctypes as C
    rng = np.random.default_rng(5)
    for (w, h, ow, oh, method) in [(48, 40, 20, 37, "bilinear"), (48, 40, 96, 38, "bilinear"), (200, 8, 100, 4, "nearest"), (33, 17, 33, 17, "bilinear")]:
        src = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
        want = np.zeros((oh, ow, 4), np.uint8)
        rc = oracle.lib.gst114_scale_4u8(C.c_void_p(src.ctypes.data), 4 * w, w, h, C.c_void_p(want.ctypes.data), 4 * ow, ow, oh,
                                         1 if method == "nearest" else 0)
        assert rc == 0
        got, _ = run(vfhip, "BGRA", w, h, src.reshape(-1), "bt601", "jpeg", method, "BGRA", ow, oh)
        assert np.array_equal(got, want)
        got, _ = run(vfhip, "BGRA", w, h, src.reshape(-1), "bt601", "jpeg", method, "RGBA", ow, oh)
        assert np.array_equal(got, want[..., [2, 1, 0, 3]])


def test_full_size_batch_properties(vfhip, oracle):
    """BASELINE config 1 shape (2160p -> 1080p), a batch of device-resident frames:
    (1) frame k of a batch == the same frame processed alone; (2) a constant frame maps to a constant frame equal
    to the oracle's single-pixel answer; (3) rows 0..63 of a random frame match the oracle run on the top slice
    (the scale is local: output row y depends on source rows 2y, 2y+1 and chroma rows y-1..y+1)."""
    import torch
    w, h, ow, oh, n = 3840, 2160, 1920, 1080, 3
    lay, size = vfhip.plane_layout("NV12", w, h)
    pitch = (size + 255) // 256 * 256
    opitch = ow * oh * 4
    g = torch.Generator(device="cpu").manual_seed(7)
    host = torch.randint(0, 256, (n, pitch), dtype=torch.uint8, generator=g)
    host[1, :size] = 77                      # constant frame
    dev_in = host.cuda()
    dev_out = torch.zeros((n, opitch), dtype=torch.uint8, device="cuda")
    cs = vfhip.ConvertScale(0)
    cs.configure("NV12", w, h, "BGRA", ow, oh, colorimetry="bt2020", chroma_site="mpeg2")
    assert cs.kernel_name == "k_cs_nv12_half"
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        cs.process_device(dev_in.data_ptr(), dev_out.data_ptr(), stream=s.cuda_stream, n_frames=n, in_pitch=pitch, out_pitch=opitch)
    s.synchronize()
    out = dev_out.cpu().numpy().reshape(n, oh, ow, 4)
    # (1)
    single = cs.process(host[2, :size].numpy()).reshape(oh, ow, 4)
    assert np.array_equal(single, out[2])
    # (2)
    const_raw = np.full(vfhip.plane_layout("NV12", 8, 8)[1], 77, np.uint8)
    px = oracle.convertscale("NV12", 8, 8, const_raw, "bt2020", "mpeg2", "bilinear", "BGRA", 4, 4)[0, 0]
    assert (out[1] == px).all()
    # (3) top slice: 136 source rows -> 68 output rows; compare the first 64 (the slice's bottom edge clamps differently)
    hs = 136
    ys = lay[0][1]
    top = np.concatenate([host[0, :ys * hs].numpy(), host[0, lay[1][0]: lay[1][0] + ys * (hs // 2)].numpy()])
    want = oracle.convertscale("NV12", w, hs, top, "bt2020", "mpeg2", "bilinear", "BGRA", ow, hs // 2)
    assert np.array_equal(out[0, :64], want[:64])
    cs.close()


@pytest.mark.parametrize("rows", [16, 8, 4])
def test_batch_grid_decomposition(vfhip, oracle, rows, monkeypatch):
    """the batched launch is one 1-D grid over (frame, strip, column group): every frame of a 16-frame batch equals
    the oracle for each strip height"""
    import torch
    w, h, ow, oh, n = 512, 136, 256, 68, 16
    monkeypatch.setenv("VFHIP_HALF_ROWS", str(rows))
    _, size = vfhip.plane_layout("NV12", w, h)
    pitch = (size + 255) // 256 * 256
    g = torch.Generator(device="cpu").manual_seed(11)
    host = torch.randint(0, 256, (n, pitch), dtype=torch.uint8, generator=g)
    dev_in, dev_out = host.cuda(), torch.zeros((n, ow * oh * 4), dtype=torch.uint8, device="cuda")
    cs = vfhip.ConvertScale(0)
    cs.configure("NV12", w, h, "RGBA", ow, oh, colorimetry="bt601", chroma_site="jpeg")
    assert cs.kernel_name == "k_cs_nv12_half"
    s = torch.cuda.Stream()
    cs.process_device(dev_in.data_ptr(), dev_out.data_ptr(), stream=s.cuda_stream, n_frames=n, in_pitch=pitch, out_pitch=ow * oh * 4)
    s.synchronize()
    out = dev_out.cpu().numpy().reshape(n, oh, ow, 4)
    for k in range(n):
        want = oracle.convertscale("NV12", w, h, host[k, :size].numpy(), "bt601", "jpeg", "bilinear", "RGBA", ow, oh)
        assert np.array_equal(out[k], want), f"frame {k}"
    cs.close()


def test_error_behaviour(vfhip):
    import ctypes as C
    cs = vfhip.ConvertScale(0)
    fi, fo = vfhip.Frame(), vfhip.Frame()
    assert vfhip.lib.vfhip_convertscale_process(cs.h, C.byref(fi), C.byref(fo)) == -3      # NOT_CONFIGURED
    with pytest.raises(vfhip.VfHipError):
        cs.configure("NV12", 0, 10, "BGRA", 10, 10)
    cs.configure("NV12", 16, 16, "BGRA", 8, 8)
    bad = vfhip.frame_from_base(vfhip.make_info("NV12", 32, 16), "NV12", 32, 16, 0)
    assert vfhip.lib.vfhip_convertscale_process(cs.h, C.byref(bad), C.byref(fo)) == -1        # caps mismatch
    cs.cleanup()
    assert vfhip.lib.vfhip_convertscale_process(cs.h, C.byref(fi), C.byref(fo)) == -3
    cs.close()


# ---- 4:2:0 outputs (staged gst-exact path: videoconvert at the input size, then per-plane videoscale) ---------------
from test_oracle_golden import MANIFEST_Y, ZY, meaningful  # noqa: E402


@pytest.mark.parametrize("case", MANIFEST_Y, ids=[c["name"] for c in MANIFEST_Y])
def test_golden_gstreamer_vectors_yuv_outputs(vfhip, case):
    c = case
    got, kname = run(vfhip, c["in_format"], c["w"], c["h"], ZY[c["name"] + "_in"], c["colorimetry"], c["chroma_site"],
                     c["method"], c["out_format"], c["ow"], c["oh"])
    assert kname == "k_cs_staged_420"
    want = ZY[c["name"] + "_out"]
    a, b = meaningful(c["out_format"], c["ow"], c["oh"], got), meaningful(c["out_format"], c["ow"], c["oh"], want)
    assert np.array_equal(a, b), f"max diff {np.abs(a.astype(int) - b.astype(int)).max()}"


@pytest.mark.parametrize("ifmt,ofmt", [("BGRA", "NV12"), ("RGBA", "I420"), ("NV12", "NV12"), ("I420", "NV12"), ("NV12", "I420"), ("I420", "I420")])
def test_yuv_outputs_1080p_vs_oracle(vfhip, oracle, ifmt, ofmt):
    """HD sizes against the oracle (1080p -> 720p and the exactly-halved 1080p -> 540p special case)"""
    rng = np.random.default_rng(17)
    for (w, h, ow, oh) in [(1920, 1080, 1280, 720), (1920, 1080, 960, 540)]:
        raw = rng.integers(0, 256, oracle_lib.raw_layout(ifmt, w, h)[1], dtype=np.uint8)
        got, _ = run(vfhip, ifmt, w, h, raw, "bt709", "mpeg2", "bilinear", ofmt, ow, oh)
        want = oracle.convertscale(ifmt, w, h, raw, "bt709", "mpeg2", "bilinear", ofmt, ow, oh)
        assert np.array_equal(meaningful(ofmt, ow, oh, got), meaningful(ofmt, ow, oh, want))


# ---- method=bicubic (videoscale method=catrom) ---------------------------------------------------------------------
from test_oracle_golden import MANIFEST_B, ZB, cubic_in_domain  # noqa: E402


@pytest.mark.parametrize("tile", [2, 1, 0], ids=["dot-tile", "float-tile", "three-pass"])
@pytest.mark.parametrize("case", [c for c in MANIFEST_B if cubic_in_domain(c)], ids=[c["name"] for c in MANIFEST_B if cubic_in_domain(c)])
def test_golden_gstreamer_vectors_bicubic(vfhip, case, tile, monkeypatch):
    """all three device paths: the tile kernel with int8 dot products (k_cs_cubic_dot, the default when a tile's windows fit its LDS planes), the
    float tile kernel (k_cs_cubic_tile) and the three-pass fallback"""
    monkeypatch.setenv("VFHIP_CUBIC_TILE", str(min(tile, 1)))
    monkeypatch.setenv("VFHIP_CUBIC_DOT", "1" if tile == 2 else "0")
    raw, want = ZB[case["name"] + "_in"], ZB[case["name"] + "_out"]
    col, site = case["colorimetry"], case["chroma_site"]
    if col is None:
        col, site = oracle_lib.default_colorimetry(case["h"])
    got, kname = run(vfhip, case["in_format"], case["w"], case["h"], raw, col, site, "bicubic", case["out_format"], case["ow"], case["oh"])
    assert kname in ("k_cs_cubic_dot", "k_cs_cubic_tile", "k_cs_ntap") and (tile or kname == "k_cs_ntap") and (tile == 2 or kname != "k_cs_cubic_dot")
    assert np.array_equal(got.reshape(-1), want), f"{kname}: {(got.reshape(-1) != want).sum()} bytes differ"


def test_bicubic_outside_the_pinned_domain_is_refused(vfhip):
    cs = vfhip.ConvertScale(0)
    for (w, h, ow, oh, ofmt, kw) in [(3, 3, 7, 5, "BGRA", {}), (16, 16, 1, 1, "BGRA", {}), (6, 36, 3, 18, "NV12", {}), (64, 36, 33, 33, "NV12", dict(add_borders=True)),
                                     (64, 36, 32, 18, "BGRA", dict(numerics="metal")),
                                     (16, 64, 32, 8, "BGRA", dict(add_borders=True))]:      # the rectangle (2 x 8) is narrower than the 32-tap filter
        with pytest.raises(vfhip.VfHipError) as e:
            cs.configure("NV12", w, h, ofmt, ow, oh, method="bicubic", **kw)
        assert e.value.code == -2                       # VFHIP_ERR_UNSUPPORTED
    cs.close()


@pytest.mark.gpu
@pytest.mark.parametrize("ifmt,w,h,ow,oh", [("I420", 54, 67, 18, 74), ("I420", 93, 32, 33, 101), ("NV12", 54, 67, 18, 74), ("NV12", 54, 67, 18, 67),
                                            ("UYVY", 60, 40, 20, 13), ("BGRA", 54, 67, 18, 74), ("NV12", 300, 200, 100, 67)])
@pytest.mark.parametrize("dot", ["1", "0"], ids=["dot-tile", "float-tile"])
def test_bicubic_alpha_where_the_taps_do_not_sum_to_64(vfhip, oracle, ifmt, w, h, ow, oh, dot, monkeypatch):
    """GStreamer's 6-bit catrom taps sum to 63 in some columns / rows at 3:1, and videoscale then outputs A = 251 for an opaque
    source (pinned: the oracle equals the real element on such vectors).  The tile kernel, which does not filter the alpha of a
    source without alpha tap by tap, has to reproduce that from the tap sums."""
    monkeypatch.setenv("VFHIP_CUBIC_DOT", dot)
    rng = np.random.default_rng(w * 1000 + h)
    raw = rng.integers(0, 256, vfhip.plane_layout(ifmt, w, h)[1], dtype=np.uint8)
    got, kname = run(vfhip, ifmt, w, h, raw, "bt601", "jpeg", "bicubic", "RGBA", ow, oh)
    want = oracle.convertscale(ifmt, w, h, raw, "bt601", "jpeg", "bicubic", "RGBA", ow, oh)
    assert np.array_equal(got, want), kname
    if ifmt != "BGRA" and (w, ow) != (300, 100):
        assert (want.reshape(oh, ow, 4)[..., 3] != 255).any()       # the case really has such columns


def test_bicubic_1080p_to_540p_vs_oracle_and_batch(vfhip, oracle):
    """the headline shape at half size (vertical pass first, 8 taps each way) against the oracle, plus a 3-frame batch"""
    import torch
    w, h, ow, oh = 1920, 1080, 960, 540
    rng = np.random.default_rng(5)
    size = vfhip.plane_layout("NV12", w, h)[1]
    frames = [rng.integers(0, 256, size, dtype=np.uint8) for _ in range(3)]
    cs = vfhip.ConvertScale(0)
    cs.configure("NV12", w, h, "BGRA", ow, oh, method="bicubic", colorimetry="bt709", chroma_site="mpeg2")
    assert cs.kernel_name == "k_cs_cubic_dot"               # 2:1 fits the tile kernel's LDS planes
    want = [oracle.convertscale("NV12", w, h, f, "bt709", "mpeg2", "bicubic", "BGRA", ow, oh) for f in frames]
    assert np.array_equal(cs.process(frames[0]).reshape(oh, ow, 4), want[0])
    pitch = (size + 255) // 256 * 256
    ring = np.zeros((3, pitch), np.uint8)
    for k, f in enumerate(frames):
        ring[k, :size] = f
    din, dout = torch.from_numpy(ring).cuda(), torch.zeros((3, ow * oh * 4), dtype=torch.uint8, device="cuda")
    s = torch.cuda.Stream()
    torch.cuda.synchronize()
    cs.process_device(din.data_ptr(), dout.data_ptr(), stream=s.cuda_stream, n_frames=3, in_pitch=pitch, out_pitch=ow * oh * 4)
    s.synchronize()
    out = dout.cpu().numpy().reshape(3, oh, ow, 4)
    for k in range(3):
        assert np.array_equal(out[k], want[k]), f"frame {k}"
    cs.close()


# ---- packed 4:2:2 inputs (UYVY / YUY2) -> RGB outputs, gst-exact -----------------------------------------------------------
from test_oracle_golden import MANIFEST_P, ZP  # noqa: E402


@pytest.mark.parametrize("case", MANIFEST_P, ids=[c["name"] for c in MANIFEST_P])
def test_golden_gstreamer_vectors_packed_inputs(vfhip, case):
    raw, want = ZP[case["name"] + "_in"], ZP[case["name"] + "_out"]
    got, kname = run(vfhip, case["in_format"], case["w"], case["h"], raw, case["colorimetry"], case["chroma_site"], case["method"],
                     case["out_format"], case["ow"], case["oh"])
    assert kname in ("k_cs_generic", "k_cs_bilinear_tile", "k_cs_cubic_dot", "k_cs_cubic_tile", "k_cs_ntap", "k_cs_uyvy_same", "k_cs_yuy2_same")            # never the metal arithmetic
    assert np.array_equal(got.reshape(-1), want), f"{kname}: {(got.reshape(-1) != want).sum()} bytes differ"


@pytest.mark.parametrize("fmt", ["UYVY", "YUY2"])
def test_packed_inputs_1080p_vs_oracle(vfhip, oracle, fmt):
    w, h, ow, oh = 1920, 1080, 1280, 720
    raw = np.random.default_rng(9).integers(0, 256, vfhip.plane_layout(fmt, w, h)[1], dtype=np.uint8)
    got, _ = run(vfhip, fmt, w, h, raw, "bt709", "mpeg2", "bilinear", "BGRA", ow, oh)
    assert np.array_equal(got, oracle.convertscale(fmt, w, h, raw, "bt709", "mpeg2", "bilinear", "BGRA", ow, oh))


# ---- I420 at exactly 2:1: k_cs_i420_half -----------------------------------------------------------------------------------
@pytest.mark.parametrize("w,h", HALF_CASES)
@pytest.mark.parametrize("col", ["bt601", "bt709", "bt2020"])
@pytest.mark.parametrize("ofmt", ["BGRA", "RGBA"])
def test_i420_half_kernel_vs_oracle(vfhip, oracle, w, h, col, ofmt):
    rng = np.random.default_rng(w * 1000 + h + 7)
    raw = rng.integers(0, 256, vfhip.plane_layout("I420", w, h)[1], dtype=np.uint8)
    got, kname = run(vfhip, "I420", w, h, raw, col, "jpeg", "bilinear", ofmt, w // 2, h // 2)
    assert kname == "k_cs_i420_half"
    want = oracle.convertscale("I420", w, h, raw, col, "jpeg", "bilinear", ofmt, w // 2, h // 2)
    assert np.array_equal(got, want), f"max diff {np.abs(got.astype(int) - want.astype(int)).max()}"


def test_i420_half_extremes_and_full_size_batch(vfhip, oracle):
    import torch
    w, h = 64, 16
    size = vfhip.plane_layout("I420", w, h)[1]
    for fill in (0, 255, None):
        raw = np.full(size, fill, np.uint8) if fill is not None else np.tile(np.array([0, 255, 255, 0, 16, 235, 240, 1], np.uint8), size // 8 + 1)[:size]
        for col in ("bt601", "bt709", "bt2020"):
            got, _ = run(vfhip, "I420", w, h, raw, col, "mpeg2", "bilinear", "BGRA", w // 2, h // 2)
            assert np.array_equal(got, oracle.convertscale("I420", w, h, raw, col, "mpeg2", "bilinear", "BGRA", w // 2, h // 2))
    # 2160p -> 1080p, a 3-frame device batch: frame 2 == alone; top slice == oracle on the slice (rows are independent here)
    w, h, ow, oh, n = 3840, 2160, 1920, 1080, 3
    lay, size = vfhip.plane_layout("I420", w, h)
    pitch = (size + 255) // 256 * 256
    g = torch.Generator(device="cpu").manual_seed(9)
    host = torch.randint(0, 256, (n, pitch), dtype=torch.uint8, generator=g)
    dev_in, dev_out = host.cuda(), torch.zeros((n, ow * oh * 4), dtype=torch.uint8, device="cuda")
    cs = vfhip.ConvertScale(0)
    cs.configure("I420", w, h, "BGRA", ow, oh, colorimetry="bt2020", chroma_site="mpeg2")
    assert cs.kernel_name == "k_cs_i420_half"
    s = torch.cuda.Stream()
    cs.process_device(dev_in.data_ptr(), dev_out.data_ptr(), stream=s.cuda_stream, n_frames=n, in_pitch=pitch, out_pitch=ow * oh * 4)
    s.synchronize()
    out = dev_out.cpu().numpy().reshape(n, oh, ow, 4)
    assert np.array_equal(cs.process(host[2, :size].numpy()).reshape(oh, ow, 4), out[2])
    hs = 128
    f0 = host[0].numpy()
    ys, cs_ = lay[0][1], lay[1][1]
    top = np.concatenate([f0[:ys * hs], f0[lay[1][0]: lay[1][0] + cs_ * (hs // 2)], f0[lay[2][0]: lay[2][0] + cs_ * (hs // 2)]])
    want = oracle.convertscale("I420", w, hs, top, "bt2020", "mpeg2", "bilinear", "BGRA", ow, hs // 2)
    assert np.array_equal(out[0, :hs // 2], want)
    cs.close()


@pytest.mark.parametrize("ifmt,ofmt,w,h,ow,oh", [("BGRA", "BGRA", 1, 9, 30, 20), ("I420", "I420", 2, 15, 116, 92), ("RGBA", "I420", 2, 28, 76, 111),
                                                 ("NV12", "BGRA", 2, 2, 64, 64), ("BGRA", "NV12", 1, 8, 9, 8)])
def test_one_sample_lines_are_replicated(vfhip, oracle, ifmt, ofmt, w, h, ow, oh):
    """a plane that is one sample wide: GStreamer's 16.16 increment formula would wrap to -1 there; both sides replicate the
    sample instead (found by tools/fuzz_gst_exact.py — the kernel used to index far outside the row)"""
    raw = np.random.default_rng(w * 100 + h).integers(0, 256, vfhip.plane_layout(ifmt, w, h)[1], dtype=np.uint8)
    got, _ = run(vfhip, ifmt, w, h, raw, "bt709", "jpeg", "bilinear", ofmt, ow, oh)
    want = oracle.convertscale(ifmt, w, h, raw, "bt709", "jpeg", "bilinear", ofmt, ow, oh)
    assert np.array_equal(np.asarray(got).reshape(-1), np.asarray(want).reshape(-1))


# ---- packed 4:2:2 outputs, packed -> 4:2:0, and the tap-quantiser ties ----------------------------------------------
from test_oracle_golden import MANIFEST_PO, ZPO, MANIFEST_T, ZT, MANIFEST_N, ZN, MANIFEST_C, ZC, gst_undefined_packed  # noqa: E402


@pytest.mark.parametrize("case", MANIFEST_PO, ids=[c["name"] for c in MANIFEST_PO])
def test_golden_gstreamer_vectors_packed_outputs(vfhip, oracle, case):
    c = case
    got, kname = run(vfhip, c["in_format"], c["w"], c["h"], ZPO[c["name"] + "_in"], c["colorimetry"], c["chroma_site"],
                     c["method"], c["out_format"], c["ow"], c["oh"])
    frames = gst_undefined_packed(oracle, c, [got, ZPO[c["name"] + "_out"]])
    if frames is None:
        assert kname == "k_cs_metal"            # outside the pinned domain: GStreamer itself emits garbage there
        return
    assert kname == ("k_cs_staged_422" if c["out_format"] in ("UYVY", "YUY2") else "k_cs_staged_420")
    a, b = (meaningful(c["out_format"], c["ow"], c["oh"], f) for f in frames)
    assert np.array_equal(a, b), f"max diff {np.abs(a.astype(int) - b.astype(int)).max()}"


@pytest.mark.parametrize("case", MANIFEST_T, ids=[c["name"] for c in MANIFEST_T])
def test_golden_gstreamer_vectors_tap_ties(vfhip, oracle, case):
    """sizes where every 2-tap weight is an exact .5 tie of the 8-bit / 6-bit quantiser (round-half-up is wrong there)"""
    c = case
    got, _ = run(vfhip, c["in_format"], c["w"], c["h"], ZT[c["name"] + "_in"], c["colorimetry"], c["chroma_site"],
                 c["method"], c["out_format"], c["ow"], c["oh"])
    want = ZT[c["name"] + "_out"]
    if c["out_format"] in ("BGRA", "RGBA"):
        assert np.array_equal(np.asarray(got).reshape(-1), want)
        return
    a, b = (meaningful(c["out_format"], c["ow"], c["oh"], f) for f in gst_undefined_packed(oracle, c, [got, want]))
    assert np.array_equal(a, b), f"max diff {np.abs(a.astype(int) - b.astype(int)).max()}"


@pytest.mark.parametrize("ifmt,ofmt", [("BGRA", "UYVY"), ("NV12", "YUY2"), ("I420", "UYVY"), ("YUY2", "UYVY"), ("UYVY", "UYVY"), ("UYVY", "NV12"), ("YUY2", "I420")])
def test_packed_outputs_hd_vs_oracle(vfhip, oracle, ifmt, ofmt):
    """HD sizes against the oracle, every byte of the frame (the oracle defines the bytes GStreamer leaves undefined)"""
    rng = np.random.default_rng(19)
    for (w, h, ow, oh) in [(1920, 1080, 1280, 720), (1279, 719, 641, 355), (640, 360, 1920, 1080)]:
        raw = rng.integers(0, 256, oracle_lib.raw_layout(ifmt, w, h)[1], dtype=np.uint8)
        got, _ = run(vfhip, ifmt, w, h, raw, "bt709", "mpeg2", "bilinear", ofmt, ow, oh)
        want = oracle.convertscale(ifmt, w, h, raw, "bt709", "mpeg2", "bilinear", ofmt, ow, oh)
        assert np.array_equal(meaningful(ofmt, ow, oh, got), meaningful(ofmt, ow, oh, want))


@pytest.mark.parametrize("case", MANIFEST_N, ids=[c["name"] for c in MANIFEST_N])
def test_golden_gstreamer_vectors_nearest_yuv_outputs(vfhip, case):
    c = case
    got, kname = run(vfhip, c["in_format"], c["w"], c["h"], ZN[c["name"] + "_in"], c["colorimetry"], c["chroma_site"],
                     c["method"], c["out_format"], c["ow"], c["oh"])
    assert kname.startswith("k_cs_staged")
    a, b = meaningful(c["out_format"], c["ow"], c["oh"], got), meaningful(c["out_format"], c["ow"], c["oh"], ZN[c["name"] + "_out"])
    assert np.array_equal(a, b), f"max diff {np.abs(a.astype(int) - b.astype(int)).max()}"


@pytest.mark.parametrize("ifmt,ofmt,w,h,ow,oh,method", [("NV12", "NV12", 200, 120, 96, 50, "bilinear"), ("BGRA", "I420", 121, 77, 64, 90, "bilinear"),
                                                        ("I420", "UYVY", 100, 60, 171, 33, "bilinear"), ("YUY2", "NV12", 90, 45, 91, 20, "nearest"),
                                                        ("UYVY", "YUY2", 64, 64, 64, 64, "bilinear"), ("RGBA", "NV12", 64, 36, 64, 36, "bilinear")])
def test_staged_cells_batched(vfhip, oracle, ifmt, ofmt, w, h, ow, oh, method):
    """the YUV-output cells take a whole batch per kernel (frame k at base + k * pitch, intermediate frames included): every
    frame of a 5-frame batch equals the oracle, and a second, larger batch re-grows the intermediate buffer"""
    import torch
    isz, osz = vfhip.plane_layout(ifmt, w, h)[1], vfhip.plane_layout(ofmt, ow, oh)[1]
    ip, op = (isz + 255) // 256 * 256, (osz + 255) // 256 * 256 + 256
    cs = vfhip.ConvertScale(0)
    cs.configure(ifmt, w, h, ofmt, ow, oh, method=method, colorimetry="bt709", chroma_site="mpeg2")
    assert cs.kernel_name.startswith("k_cs_staged")
    s = torch.cuda.Stream()
    for n in (5, 9):
        g = torch.Generator(device="cpu").manual_seed(100 + n)
        host = torch.randint(0, 256, (n, ip), dtype=torch.uint8, generator=g)
        dev_in, dev_out = host.cuda(), torch.zeros((n, op), dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        cs.process_device(dev_in.data_ptr(), dev_out.data_ptr(), stream=s.cuda_stream, n_frames=n, in_pitch=ip, out_pitch=op)
        s.synchronize()
        out = dev_out.cpu().numpy()
        for k in range(n):
            want = oracle.convertscale(ifmt, w, h, host[k, :isz].numpy(), "bt709", "mpeg2", method, ofmt, ow, oh)
            assert np.array_equal(meaningful(ofmt, ow, oh, out[k, :osz]), meaningful(ofmt, ow, oh, want)), f"batch {n} frame {k}"
    cs.close()


def test_staged_scalar_fallbacks_match(vfhip, oracle, monkeypatch):
    """the byte-wise kernels that take over when rows are not 4-byte aligned (forced here with the tuning knobs) produce the
    same frames as the dword / v_dot4 variants"""
    monkeypatch.setenv("VFHIP_RGB2YUV_SCALAR", "1")
    monkeypatch.setenv("VFHIP_PLANE_SCALAR", "1")
    rng = np.random.default_rng(77)
    for (ifmt, ofmt, w, h, ow, oh) in [("BGRA", "NV12", 64, 36, 64, 36), ("RGBA", "I420", 67, 41, 33, 20), ("NV12", "NV12", 128, 72, 64, 36),
                                       ("I420", "I420", 128, 72, 64, 72), ("NV12", "I420", 96, 54, 96, 27), ("NV12", "UYVY", 67, 41, 67, 41),
                                       ("NV12", "YUY2", 64, 36, 50, 20), ("YUY2", "NV12", 67, 41, 67, 41), ("UYVY", "NV12", 64, 36, 90, 50),
                                       ("BGRA", "UYVY", 67, 41, 67, 41), ("RGBA", "YUY2", 64, 36, 32, 18)]:
        raw = rng.integers(0, 256, oracle_lib.raw_layout(ifmt, w, h)[1], dtype=np.uint8)
        got, _ = run(vfhip, ifmt, w, h, raw, "bt709", "mpeg2", "bilinear", ofmt, ow, oh)
        want = oracle.convertscale(ifmt, w, h, raw, "bt709", "mpeg2", "bilinear", ofmt, ow, oh)
        assert np.array_equal(meaningful(ofmt, ow, oh, got), meaningful(ofmt, ow, oh, want)), (ifmt, ofmt, w, h, ow, oh)


@pytest.mark.parametrize("case", MANIFEST_C, ids=[c["name"] for c in MANIFEST_C])
def test_golden_gstreamer_vectors_bicubic_yuv_outputs(vfhip, oracle, case):
    """videoscale method=catrom with YUV outputs: catrom on the luma plane / the lines of a packed frame, un-limited LINEAR taps
    on the chroma planes of a planar frame"""
    c = case
    got, kname = run(vfhip, c["in_format"], c["w"], c["h"], ZC[c["name"] + "_in"], c["colorimetry"], c["chroma_site"],
                     c["method"], c["out_format"], c["ow"], c["oh"])
    assert kname.startswith("k_cs_staged")
    a, b = (meaningful(c["out_format"], c["ow"], c["oh"], f) for f in gst_undefined_packed(oracle, c, [got, ZC[c["name"] + "_out"]]))
    assert np.array_equal(a, b), f"max diff {np.abs(a.astype(int) - b.astype(int)).max()}"


def test_bicubic_yuv_outputs_hd_vs_oracle(vfhip, oracle):
    rng = np.random.default_rng(23)
    for (ifmt, ofmt, w, h, ow, oh) in [("NV12", "NV12", 1920, 1080, 1280, 720), ("BGRA", "I420", 1280, 720, 1920, 1080), ("UYVY", "UYVY", 1280, 720, 640, 360),
                                       ("I420", "YUY2", 1279, 719, 641, 355)]:
        raw = rng.integers(0, 256, oracle_lib.raw_layout(ifmt, w, h)[1], dtype=np.uint8)
        got, _ = run(vfhip, ifmt, w, h, raw, "bt709", "mpeg2", "bicubic", ofmt, ow, oh)
        want = oracle.convertscale(ifmt, w, h, raw, "bt709", "mpeg2", "bicubic", ofmt, ow, oh)
        assert np.array_equal(meaningful(ofmt, ow, oh, got), meaningful(ofmt, ow, oh, want)), (ifmt, ofmt)


def _custom_layout(vfhip, fmt, w, h, pad, base):
    """plane list [(offset, stride, rows)] with every stride padded by `pad` bytes and the first plane at byte `base`"""
    pl, _ = vfhip.plane_layout(fmt, w, h)
    out, off = [], base
    for (_, stride, rows) in pl:
        out.append((off, stride + pad, rows))
        off += (stride + pad) * rows + 5                      # planes need not follow each other directly
    return out, off


def _repack(buf, layout_from, layout_to, size_to):
    """copy plane rows between two layouts of the same format and size (the narrower stride bounds the copy)"""
    out = np.zeros(size_to, np.uint8)
    for (fo, fs, rows), (to, ts, _) in zip(layout_from, layout_to):
        n = min(fs, ts)
        for r in range(rows):
            out[to + r * ts: to + r * ts + n] = buf[fo + r * fs: fo + r * fs + n]
    return out


@pytest.mark.parametrize("pad,base", [(16, 0), (3, 0), (16, 1), (5, 3), (64, 8)])
@pytest.mark.parametrize("ifmt,ofmt,w,h,ow,oh,method", [("NV12", "BGRA", 128, 72, 64, 36, "bilinear"), ("I420", "RGBA", 128, 72, 64, 36, "bilinear"), ("NV12", "BGRA", 96, 54, 50, 31, "bilinear"),
                                                        ("NV12", "RGBA", 80, 46, 100, 60, "nearest"), ("NV12", "BGRA", 96, 54, 40, 30, "bicubic"), ("BGRA", "NV12", 64, 36, 64, 36, "bilinear"),
                                                        ("RGBA", "I420", 66, 38, 33, 19, "bilinear"), ("NV12", "NV12", 128, 72, 64, 36, "bilinear"), ("I420", "NV12", 90, 50, 60, 40, "bilinear"),
                                                        ("NV12", "UYVY", 64, 36, 64, 36, "bilinear"), ("YUY2", "NV12", 64, 36, 48, 30, "bilinear"), ("UYVY", "YUY2", 70, 40, 35, 20, "bicubic"),
                                                        ("BGRA", "BGRA", 64, 36, 100, 50, "bilinear"), ("UYVY", "BGRA", 64, 36, 32, 18, "bilinear")])
def test_padded_strides_and_misaligned_planes(vfhip, oracle, ifmt, ofmt, w, h, ow, oh, method, pad, base):
    """frames as decoders and pools hand them over: padded strides, gaps between planes, odd strides and base addresses (which
    take the byte-wise kernel variants) - the same bytes as the default layout in every case"""
    import torch
    (ipl, isz), (opl, osz) = _custom_layout(vfhip, ifmt, w, h, pad, base), _custom_layout(vfhip, ofmt, ow, oh, pad, base)
    dpl_in, dsz_in = vfhip.plane_layout(ifmt, w, h)
    dpl_out, dsz_out = vfhip.plane_layout(ofmt, ow, oh)
    raw = np.random.default_rng(w * 7 + pad).integers(0, 256, dsz_in, dtype=np.uint8)
    padded = _repack(raw, dpl_in, ipl, isz + 64)
    din = torch.from_numpy(padded).cuda()
    dout = torch.zeros(osz + 64, dtype=torch.uint8, device="cuda")
    cs = vfhip.ConvertScale(0)
    cs.configure(ifmt, w, h, ofmt, ow, oh, method=method, colorimetry="bt709", chroma_site="mpeg2")
    s = torch.cuda.Stream()
    torch.cuda.synchronize()
    cs.process_device(din.data_ptr(), dout.data_ptr(), stream=s.cuda_stream, in_layout=(ipl, isz), out_layout=(opl, osz))
    s.synchronize()
    cs.close()
    got = _repack(dout.cpu().numpy(), opl, dpl_out, dsz_out)
    want = np.asarray(oracle.convertscale(ifmt, w, h, raw, "bt709", "mpeg2", method, ofmt, ow, oh)).reshape(-1)
    if ofmt in ("BGRA", "RGBA"):
        assert np.array_equal(got, want)
    else:
        assert np.array_equal(meaningful(ofmt, ow, oh, got), meaningful(ofmt, ow, oh, want))


@pytest.mark.parametrize("pad,base", [(16, 0), (3, 1)])
@pytest.mark.parametrize("ifmt,ofmt,w,h,ow,oh", [("NV12", "BGRA", 128, 72, 64, 36), ("I420", "NV12", 90, 50, 60, 40), ("BGRA", "UYVY", 64, 36, 33, 20)])
def test_host_frames_with_padded_strides(vfhip, oracle, ifmt, ofmt, w, h, ow, oh, pad, base):
    """the host-frame entry point (staging upload / download) honours per-plane strides and offsets too"""
    import ctypes as C
    (ipl, isz), (opl, osz) = _custom_layout(vfhip, ifmt, w, h, pad, base), _custom_layout(vfhip, ofmt, ow, oh, pad, base)
    dpl_in, dsz_in = vfhip.plane_layout(ifmt, w, h)
    dpl_out, dsz_out = vfhip.plane_layout(ofmt, ow, oh)
    raw = np.random.default_rng(w + pad).integers(0, 256, dsz_in, dtype=np.uint8)
    hin, hout = _repack(raw, dpl_in, ipl, isz + 64), np.full(osz + 64, 0xAB, np.uint8)
    cs = vfhip.ConvertScale(0)
    cs.configure(ifmt, w, h, ofmt, ow, oh, colorimetry="bt709", chroma_site="mpeg2")
    fi = vfhip.frame_from_base(cs.in_info, ifmt, w, h, hin.ctypes.data, layout=(ipl, isz))
    fo = vfhip.frame_from_base(cs.out_info, ofmt, ow, oh, hout.ctypes.data, layout=(opl, osz))
    vfhip.check(vfhip.lib.vfhip_convertscale_process(cs.h, C.byref(fi), C.byref(fo)))
    cs.close()
    got = _repack(hout, opl, dpl_out, dsz_out)
    want = np.asarray(oracle.convertscale(ifmt, w, h, raw, "bt709", "mpeg2", "bilinear", ofmt, ow, oh)).reshape(-1)
    if ofmt in ("BGRA", "RGBA"):
        assert np.array_equal(got, want)
    else:
        assert np.array_equal(meaningful(ofmt, ow, oh, got), meaningful(ofmt, ow, oh, want))
    # bytes outside the planes' rows are not the library's to write: the gaps between planes keep the fill pattern
    for k in range(len(opl) - 1):
        end = opl[k][0] + opl[k][1] * (opl[k][2] if k else oh)
        assert (hout[opl[k + 1][0] - 5: opl[k + 1][0]] == 0xAB).all() or end > opl[k + 1][0] - 5
    assert (hout[-32:] == 0xAB).all()


@pytest.mark.parametrize("ifmt,ofmt,w,h,ow,oh,method", [("NV12", "BGRA", 64, 36, 48, 48, "bilinear"), ("I420", "RGBA", 36, 64, 50, 40, "bilinear"), ("BGRA", "BGRA", 40, 30, 64, 64, "nearest"),
                                                        ("UYVY", "RGBA", 64, 16, 33, 31, "bilinear"), ("NV12", "BGRA", 128, 72, 64, 36, "bilinear"),
                                                        ("NV12", "BGRA", 64, 36, 48, 48, "bicubic"), ("I420", "RGBA", 200, 120, 97, 120, "bicubic"), ("BGRA", "BGRA", 40, 30, 64, 64, "bicubic")])
def test_gst_exact_letterbox(vfhip, oracle, ifmt, ofmt, w, h, ow, oh, method):
    """add-borders with gst-exact numerics and an RGB output: the reference's centred aspect-preserving rectangle
    (metalconvertscalerenderer.m:137-166) holds exactly what `videoconvert ! videoscale` gives at the rectangle's size, the
    rest is the border colour in the output's byte order"""
    import math
    raw = np.random.default_rng(ow).integers(0, 256, vfhip.plane_layout(ifmt, w, h)[1], dtype=np.uint8)
    cs = vfhip.ConvertScale(0)
    cs.configure(ifmt, w, h, ofmt, ow, oh, method=method, add_borders=True, border_color=0x80FF2010, colorimetry="bt709", chroma_site="mpeg2")
    assert cs.kernel_name != "k_cs_metal"
    got = cs.process(raw).reshape(oh, ow, 4)
    cs.close()
    src, dst = np.float32(w) / np.float32(h), np.float32(ow) / np.float32(oh)
    rw, rh = ow, oh
    if src > dst:
        rh = int(math.floor(float(oh) * float(np.float32(dst / src)) + 0.5))
    else:
        rw = int(math.floor(float(ow) * float(np.float32(src / dst)) + 0.5))
    rw, rh = min(max(rw, 1), ow), min(max(rh, 1), oh)
    rx, ry = (ow - rw) // 2, (oh - rh) // 2
    inner = oracle.convertscale(ifmt, w, h, raw, "bt709", "mpeg2", method, ofmt, rw, rh)
    assert np.array_equal(got[ry:ry + rh, rx:rx + rw], inner)
    a, r, g, b = 0x80, 0xFF, 0x20, 0x10
    colour = np.array([r, g, b, a] if ofmt == "RGBA" else [b, g, r, a], np.uint8)
    mask = np.ones((oh, ow), bool)
    mask[ry:ry + rh, rx:rx + rw] = False
    assert (got[mask] == colour).all() and (rw < ow or rh < oh or not mask.any())


def test_bicubic_yuv_composed_and_two_pass_forms_agree(vfhip, oracle, monkeypatch):
    """method=bicubic on planes runs as two kernels through an intermediate plane; the composed one-kernel form (kept for packed
    frames and mixed 2-tap / n-tap planes, forced here) gives the same bytes"""
    rng = np.random.default_rng(29)
    cases = [("NV12", "NV12", 200, 120, 96, 50), ("I420", "I420", 96, 54, 200, 120), ("BGRA", "NV12", 121, 77, 64, 90), ("NV12", "I420", 130, 200, 90, 60)]
    for composed in (False, True):
        if composed:
            monkeypatch.setenv("VFHIP_PLANE_COMPOSED", "1")
        for (ifmt, ofmt, w, h, ow, oh) in cases:
            raw = np.random.default_rng(w).integers(0, 256, oracle_lib.raw_layout(ifmt, w, h)[1], dtype=np.uint8)
            got, _ = run(vfhip, ifmt, w, h, raw, "bt709", "mpeg2", "bicubic", ofmt, ow, oh)
            want = oracle.convertscale(ifmt, w, h, raw, "bt709", "mpeg2", "bicubic", ofmt, ow, oh)
            assert np.array_equal(meaningful(ofmt, ow, oh, got), meaningful(ofmt, ow, oh, want)), (composed, ifmt, ofmt)
    del rng


@pytest.mark.parametrize("ifmt,ofmt,ow,oh", [("NV12", "BGRA", 3840, 2160), ("NV12", "NV12", 3840, 2160), ("I420", "RGBA", 2731, 1537)])
def test_8k_frames(vfhip, oracle, ifmt, ofmt, ow, oh):
    """7680 x 4320 inputs (the largest size the reference's caps template allows is 8192): index arithmetic and grids at 8K"""
    w, h = 7680, 4320
    raw = np.random.default_rng(8).integers(0, 256, oracle_lib.raw_layout(ifmt, w, h)[1], dtype=np.uint8)
    got, _ = run(vfhip, ifmt, w, h, raw, "bt2020", "mpeg2", "bilinear", ofmt, ow, oh)
    want = oracle.convertscale(ifmt, w, h, raw, "bt2020", "mpeg2", "bilinear", ofmt, ow, oh)
    if ofmt in ("BGRA", "RGBA"):
        assert np.array_equal(got, want)
    else:
        assert np.array_equal(meaningful(ofmt, ow, oh, got), meaningful(ofmt, ow, oh, want))


@pytest.mark.parametrize("ifmt,ofmt,w,h,ow,oh,method", [("NV12", "NV12", 64, 36, 64, 64, "bilinear"), ("BGRA", "I420", 36, 64, 64, 64, "bilinear"), ("I420", "UYVY", 64, 16, 32, 31, "nearest"),
                                                        ("UYVY", "NV12", 128, 72, 64, 64, "bilinear"), ("NV12", "YUY2", 96, 54, 64, 64, "bicubic"), ("RGBA", "NV12", 40, 30, 64, 64, "bilinear")])
def test_gst_exact_letterbox_yuv_outputs(vfhip, oracle, ifmt, ofmt, w, h, ow, oh, method):
    """add-borders with gst-exact numerics and a YUV output: the reference's centred rectangle holds exactly what the two-step
    path gives at the rectangle's size, every other sample is the border colour through the RGB -> YUV matrix (rectangles on
    chroma-sample boundaries; others run the metal arithmetic)"""
    import math
    raw = np.random.default_rng(ow + oh).integers(0, 256, vfhip.plane_layout(ifmt, w, h)[1], dtype=np.uint8)
    cs = vfhip.ConvertScale(0)
    cs.configure(ifmt, w, h, ofmt, ow, oh, method=method, add_borders=True, border_color=0xFF2060C0, colorimetry="bt601", chroma_site="jpeg")
    assert cs.kernel_name.startswith("k_cs_staged")
    got = cs.process(raw)
    cs.close()
    src, dst = np.float32(w) / np.float32(h), np.float32(ow) / np.float32(oh)
    rw, rh = ow, oh
    if src > dst:
        rh = int(math.floor(float(oh) * float(np.float32(dst / src)) + 0.5))
    else:
        rw = int(math.floor(float(ow) * float(np.float32(src / dst)) + 0.5))
    rx, ry = (ow - rw) // 2, (oh - rh) // 2
    assert not (rx | rw) & 1 and (ofmt in ("UYVY", "YUY2") or not (ry | rh) & 1), "pick a case whose rectangle is aligned"
    inner = np.asarray(oracle.convertscale(ifmt, w, h, raw, "bt601", "jpeg", method, ofmt, rw, rh))
    r, g, b = 0x20, 0x60, 0xC0
    Y, U, V = ((66 * r + 129 * g + 25 * b) >> 8) + 16, ((-38 * r - 74 * g + 112 * b) >> 8) + 128, ((112 * r - 94 * g - 18 * b) >> 8) + 128
    opl, _ = vfhip.plane_layout(ofmt, ow, oh)
    ipl, _ = vfhip.plane_layout(ofmt, rw, rh)
    want = np.zeros_like(got)
    if ofmt in ("UYVY", "YUY2"):
        (o0, os_, _), (i0, is_, _) = opl[0], ipl[0]
        rows = want[o0: o0 + os_ * oh].reshape(oh, os_)
        mp = [Y, U, Y, V] if ofmt == "YUY2" else [U, Y, V, Y]
        rows[:, : 4 * ((ow + 1) // 2)] = np.tile(np.array(mp, np.uint8), (ow + 1) // 2)
        rows[ry: ry + rh, 2 * rx: 2 * rx + 2 * rw] = inner[i0: i0 + is_ * rh].reshape(rh, is_)[:, : 2 * rw]
    else:
        for k, ((oo, os_, orows), (io, is_, irows)) in enumerate(zip(opl, ipl)):
            sub = 1 if k == 0 else 2
            n = 2 if (ofmt == "NV12" and k == 1) else 1
            pw, ph = (ow if k == 0 else (ow + 1) // 2), (oh if k == 0 else (oh + 1) // 2)
            plane = want[oo: oo + os_ * orows].reshape(orows, os_)
            if k == 0:
                plane[:ph, :pw] = Y
            elif ofmt == "NV12":
                plane[:ph, 0: 2 * pw: 2] = U
                plane[:ph, 1: 2 * pw: 2] = V
            else:
                plane[:ph, :pw] = U if k == 1 else V
            iw_, ih_ = (rw if k == 0 else rw // 2), (rh if k == 0 else rh // 2)
            plane[ry // sub: ry // sub + ih_, n * (rx // sub): n * (rx // sub) + n * iw_] = inner[io: io + is_ * irows].reshape(irows, is_)[:ih_, : n * iw_]
    assert np.array_equal(meaningful(ofmt, ow, oh, got), meaningful(ofmt, ow, oh, want))


def test_gst_exact_letterbox_yuv_batched(vfhip):
    """the bordered YUV output as a batch on device frames equals the one-frame host call, frame by frame"""
    import torch
    w, h, ow, oh, n = 64, 36, 64, 64, 4
    isz, osz = vfhip.plane_layout("NV12", w, h)[1], vfhip.plane_layout("I420", ow, oh)[1]
    ip, op = (isz + 255) // 256 * 256, (osz + 255) // 256 * 256
    cs = vfhip.ConvertScale(0)
    cs.configure("NV12", w, h, "I420", ow, oh, add_borders=True, border_color=0xFF804020, colorimetry="bt709", chroma_site="mpeg2")
    assert cs.kernel_name == "k_cs_staged_420"
    host = torch.randint(0, 256, (n, ip), dtype=torch.uint8, generator=torch.Generator().manual_seed(3))
    want = [cs.process(host[k, :isz].numpy()) for k in range(n)]
    dev_in, dev_out = host.cuda(), torch.zeros((n, op), dtype=torch.uint8, device="cuda")
    s = torch.cuda.Stream()
    torch.cuda.synchronize()
    cs.process_device(dev_in.data_ptr(), dev_out.data_ptr(), stream=s.cuda_stream, n_frames=n, in_pitch=ip, out_pitch=op)
    s.synchronize()
    out = dev_out.cpu().numpy()
    for k in range(n):
        assert np.array_equal(meaningful("I420", ow, oh, out[k, :osz]), meaningful("I420", ow, oh, want[k])), f"frame {k}"
    cs.close()


from test_oracle_golden import MANIFEST_MS, ZMS  # noqa: E402


@pytest.mark.parametrize("case", MANIFEST_MS, ids=[c["name"] for c in MANIFEST_MS])
def test_golden_gstreamer_vectors_mixed_sitings(vfhip, oracle, case):
    """different chroma sitings on the two sides of a YUV -> YUV cell (NV12 <-> packed resample with both, the I420 <-> packed
    fast paths and the UYVY <-> YUY2 swizzle ignore them)"""
    c = case
    cs = vfhip.ConvertScale(0)
    cs.configure(c["in_format"], c["w"], c["h"], c["out_format"], c["ow"], c["oh"], colorimetry=c["colorimetry"], chroma_site=c["chroma_site"],
                 out_chroma_site=c["out_chroma_site"])
    assert cs.kernel_name.startswith("k_cs_staged")
    got = cs.process(ZMS[c["name"] + "_in"])
    cs.close()
    a, b = (meaningful(c["out_format"], c["ow"], c["oh"], f) for f in gst_undefined_packed(oracle, c, [got, ZMS[c["name"] + "_out"]]))
    assert np.array_equal(a, b)


from test_oracle_golden import MANIFEST_RM, ZRM  # noqa: E402


@pytest.mark.parametrize("case", MANIFEST_RM, ids=[c["name"] for c in MANIFEST_RM])
def test_golden_gstreamer_vectors_matrix_and_siting_changes(vfhip, oracle, case):
    """YUV -> YUV with a colour-matrix change (NV12 / I420 / UYVY / YUY2 either side) and NV12 <-> I420 with a siting change:
    videoconvert's generic path (k_yuv_to_yuv) + the usual scale stage, byte for byte against the real GStreamer 1.14 pipeline"""
    c = case
    cs = vfhip.ConvertScale(0)
    cs.configure(c["in_format"], c["w"], c["h"], c["out_format"], c["ow"], c["oh"], colorimetry=c["colorimetry"], chroma_site=c["chroma_site"],
                 out_chroma_site=c["out_chroma_site"], out_colorimetry=c["out_colorimetry"], numerics="gst-exact-strict")
    assert cs.kernel_name.startswith("k_cs_staged") and cs.numerics_in_effect == "gst-exact"
    got = cs.process(ZRM[c["name"] + "_in"])
    cs.close()
    fr = gst_undefined_packed(oracle, c, [got, ZRM[c["name"] + "_out"]])
    if fr is None:
        pytest.skip("GStreamer 1.14 emits out-of-line garbage for this packed frame")
    a, b = (meaningful(c["out_format"], c["ow"], c["oh"], f) for f in fr)
    assert np.array_equal(a, b)


@pytest.mark.parametrize("ifmt,ofmt,w,h,ow,oh,method", [("NV12", "NV12", 1920, 1080, 1280, 720, "bilinear"), ("I420", "NV12", 1280, 720, 1280, 720, "bilinear"),
                                                        ("UYVY", "I420", 642, 361, 320, 181, "nearest"), ("NV12", "YUY2", 640, 360, 960, 540, "bicubic")])
def test_matrix_change_hd_vs_oracle_and_batch(vfhip, oracle, ifmt, ofmt, w, h, ow, oh, method):
    """bigger frames, the other scale methods and the batched device entry point (2 frames) of the matrix-change cells vs the oracle"""
    import torch
    rng = np.random.default_rng(w + h)
    isz, osz = oracle_lib.raw_layout(ifmt, w, h)[1], oracle_lib.raw_layout(ofmt, ow, oh)[1]
    frames = [rng.integers(0, 256, isz, dtype=np.uint8) for _ in range(2)]
    cs = vfhip.ConvertScale(0)
    cs.configure(ifmt, w, h, ofmt, ow, oh, method=method, colorimetry="bt709", chroma_site="mpeg2", out_colorimetry="bt601", out_chroma_site="jpeg", numerics="gst-exact-strict")
    ip, op = (isz + 255) // 256 * 256, (osz + 255) // 256 * 256
    din = torch.zeros((2, ip), dtype=torch.uint8, device="cuda")
    for k in range(2):
        din[k, :isz] = torch.from_numpy(frames[k]).cuda()
    dout = torch.zeros((2, op), dtype=torch.uint8, device="cuda")
    s = torch.cuda.Stream()
    torch.cuda.synchronize()
    cs.process_device(din.data_ptr(), dout.data_ptr(), stream=s.cuda_stream, n_frames=2, in_pitch=ip, out_pitch=op)
    s.synchronize()
    out = dout.cpu().numpy()
    for k in range(2):
        want = oracle.convertscale(ifmt, w, h, frames[k], "bt709", "mpeg2", method, ofmt, ow, oh, out_chroma_site="jpeg", out_colorimetry="bt601")
        assert np.array_equal(meaningful(ofmt, ow, oh, out[k, :osz]), meaningful(ofmt, ow, oh, want)), f"frame {k}"
        assert np.array_equal(meaningful(ofmt, ow, oh, cs.process(frames[k])), meaningful(ofmt, ow, oh, want))
    cs.close()


@pytest.mark.gpu
@pytest.mark.parametrize("ifmt", ["NV12", "I420", "UYVY", "YUY2"])
@pytest.mark.parametrize("w,h", [(64, 36), (200, 113), (16, 3), (1920, 1080), (24, 2), (136, 77)])
@pytest.mark.parametrize("col,site", [("bt601", "jpeg"), ("bt709", "mpeg2"), ("bt2020", "mpeg2")])
def test_yuv420_same_size_conversion(vfhip, oracle, ifmt, w, h, col, site, monkeypatch):
    """the element as a plain converter: NV12 / I420 -> BGRA / RGBA at the same size runs k_cs_yuv_same (eight pixels per lane) — against
    the oracle, against the generic kernel it replaces, for both scaling methods (videoscale passes through either way), odd heights included"""
    rng = np.random.default_rng(w * 7 + h)
    raw = rng.integers(0, 256, vfhip.plane_layout(ifmt, w, h)[1], dtype=np.uint8)
    for ofmt in ("BGRA", "RGBA"):
        want = oracle.convertscale(ifmt, w, h, raw, col, site, "bilinear", ofmt, w, h)
        for method in ("bilinear", "nearest"):
            got, kname = run(vfhip, ifmt, w, h, raw, col, site, method, ofmt, w, h)
            assert kname == f"k_cs_{ifmt.lower()}_same", kname
            assert np.array_equal(got, want), (ofmt, method)
        monkeypatch.setenv("VFHIP_NO_SAME", "1")
        tile, kname = run(vfhip, ifmt, w, h, raw, col, site, "bilinear", ofmt, w, h)
        assert kname == "k_cs_bilinear_tile" and np.array_equal(tile, want)
        monkeypatch.setenv("VFHIP_NO_BILINEAR_TILE", "1")
        old, kname = run(vfhip, ifmt, w, h, raw, col, site, "bilinear", ofmt, w, h)
        monkeypatch.delenv("VFHIP_NO_BILINEAR_TILE")
        monkeypatch.delenv("VFHIP_NO_SAME")
        assert kname == ("k_cs_taps" if ifmt in ("NV12", "I420") and w >= 8 else "k_cs_generic") and np.array_equal(old, want)


@pytest.mark.gpu
def test_nv12_same_size_batch_and_unaligned_fallback(vfhip, oracle):
    """a batch through k_cs_nv12_same on the device path, and a frame whose planes miss the kernel's alignment contract (the generic
    kernel must take over for that call and produce the same bytes)"""
    import torch
    w, h, n = 320, 90, 3
    rng = np.random.default_rng(8)
    size = vfhip.plane_layout("NV12", w, h)[1]
    frames = [rng.integers(0, 256, size, dtype=np.uint8) for _ in range(n)]
    want = [oracle.convertscale("NV12", w, h, f, "bt709", "mpeg2", "bilinear", "BGRA", w, h) for f in frames]
    cs = vfhip.ConvertScale(0)
    cs.configure("NV12", w, h, "BGRA", w, h, colorimetry="bt709", chroma_site="mpeg2")
    assert cs.kernel_name == "k_cs_nv12_same"
    pitch = (size + 255) // 256 * 256
    ring = np.zeros((n, pitch), np.uint8)
    for k, f in enumerate(frames):
        ring[k, :size] = f
    din, dout = torch.from_numpy(ring).cuda(), torch.zeros((n, w * h * 4), dtype=torch.uint8, device="cuda")
    s = torch.cuda.Stream()
    torch.cuda.synchronize()
    cs.process_device(din.data_ptr(), dout.data_ptr(), stream=s.cuda_stream, n_frames=n, in_pitch=pitch, out_pitch=w * h * 4)
    s.synchronize()
    out = dout.cpu().numpy()
    for k in range(n):
        assert np.array_equal(out[k].reshape(h, w, 4), want[k].reshape(h, w, 4)), k
    # the input 4 bytes off an 8-byte boundary
    flat = torch.zeros(size + 64, dtype=torch.uint8, device="cuda")
    flat[4:4 + size] = torch.from_numpy(frames[0]).cuda()
    one = torch.zeros(w * h * 4, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    cs.process_device(flat.data_ptr() + 4, one.data_ptr(), stream=s.cuda_stream)
    s.synchronize()
    assert np.array_equal(one.cpu().numpy().reshape(h, w, 4), want[0].reshape(h, w, 4))
    cs.close()


@pytest.mark.gpu
@pytest.mark.parametrize("ifmt", ["NV12", "I420", "BGRA", "UYVY"])
@pytest.mark.parametrize("w,h,ow,oh", [(192, 108, 384, 216), (192, 108, 128, 72), (200, 113, 333, 77), (64, 36, 640, 360), (1920, 1080, 1280, 720),
                                       (130, 70, 61, 200), (96, 54, 96, 120), (96, 54, 150, 54), (17, 9, 40, 31), (640, 360, 1000, 700), (100, 100, 101, 100)])
def test_bilinear_tile_against_oracle_and_per_pixel_kernels(vfhip, oracle, ifmt, w, h, ow, oh, monkeypatch):
    """k_cs_bilinear_tile (conversion once per tile into LDS, then GStreamer's two 2-tap passes in its order) — up-scales, down-scales,
    one axis only, both pass orders, odd sizes — against the oracle and against the per-pixel kernel it replaces"""
    rng = np.random.default_rng(w * 31 + oh)
    raw = rng.integers(0, 256, vfhip.plane_layout(ifmt, w, h)[1], dtype=np.uint8)
    want = oracle.convertscale(ifmt, w, h, raw, "bt709", "mpeg2", "bilinear", "RGBA", ow, oh)
    got, kname = run(vfhip, ifmt, w, h, raw, "bt709", "mpeg2", "bilinear", "RGBA", ow, oh)
    if ow >= w and oh >= h:
        assert kname == "k_cs_bilinear_tile", kname             # no minification: always the tile kernel
    elif ifmt in ("BGRA", "I420") or w < 16:
        assert kname in ("k_cs_taps", "k_cs_generic"), kname    # down-scales of RGB and I420 inputs stay per pixel
    elif ifmt == "NV12" and ow < w and oh < h:
        assert kname == "k_cs_taps", kname                      # NV12 minified on both axes: k_cs_taps (in strips when the launch is large enough)
    else:
        assert kname in ("k_cs_bilinear_tile", "k_cs_taps", "k_cs_generic"), kname      # YUV inputs: the tile kernel while a tile's source region fits its LDS arrays
        if w / ow <= 1.6 and h / oh <= 1.6:
            assert kname == "k_cs_bilinear_tile", kname
    assert np.array_equal(got, want)
    monkeypatch.setenv("VFHIP_NO_BILINEAR_TILE", "1")
    old, kname = run(vfhip, ifmt, w, h, raw, "bt709", "mpeg2", "bilinear", "RGBA", ow, oh)
    monkeypatch.delenv("VFHIP_NO_BILINEAR_TILE")
    assert kname in ("k_cs_taps", "k_cs_generic") and np.array_equal(old, want)


@pytest.mark.gpu
@pytest.mark.parametrize("ifmt", ["NV12", "I420"])
@pytest.mark.parametrize("site", ["mpeg2", "jpeg"])
@pytest.mark.parametrize("w,h,ow,oh", [(192, 108, 128, 72), (200, 113, 67, 51), (1920, 1080, 640, 480), (130, 70, 61, 69), (130, 70, 129, 31), (64, 36, 21, 9), (96, 54, 95, 53),
                                       (320, 180, 100, 179), (258, 258, 65, 65), (640, 360, 213, 120), (18, 10, 9, 5), (100, 64, 99, 200), (8, 8, 3, 3), (1000, 30, 64, 7)])
def test_taps_strip_kernel(vfhip, oracle, ifmt, site, w, h, ow, oh, monkeypatch):
    """k_cs_taps_strip (four output rows per lane, the two source rows' shared chroma rows fetched and filtered once, NV12 and I420 windows) at sizes
    that leave partial strips, partial waves, clamped edge rows and both pass orders — forced on whatever the launch size — against the oracle
    and against k_cs_taps at one row per lane"""
    rng = np.random.default_rng(w * 13 + oh)
    raw = rng.integers(0, 256, vfhip.plane_layout(ifmt, w, h)[1], dtype=np.uint8)
    col = "bt601" if site == "jpeg" else "bt709"
    for ofmt in ("BGRA", "RGBA"):
        want = oracle.convertscale(ifmt, w, h, raw, col, site, "bilinear", ofmt, ow, oh)
        monkeypatch.setenv("VFHIP_TAPS_FILL", "0")
        got, kname = run(vfhip, ifmt, w, h, raw, col, site, "bilinear", ofmt, ow, oh)
        assert kname in ("k_cs_taps", "k_cs_bilinear_tile"), kname
        assert np.array_equal(got, want), (ofmt, "strips")
        monkeypatch.setenv("VFHIP_TAPS_ROWS", "1")
        one, kname1 = run(vfhip, ifmt, w, h, raw, col, site, "bilinear", ofmt, ow, oh)
        monkeypatch.delenv("VFHIP_TAPS_ROWS")
        monkeypatch.delenv("VFHIP_TAPS_FILL")
        assert kname1 == kname and np.array_equal(one, want), (ofmt, "one row per lane")


@pytest.mark.gpu
@pytest.mark.parametrize("ifmt,ofmt", [("NV12", "I420"), ("I420", "NV12")])
@pytest.mark.parametrize("w,h,ow,oh", [(64, 36, 64, 36), (1920, 1080, 1920, 1080), (32, 2, 32, 2), (16, 4, 16, 4), (40, 20, 40, 20), (64, 37, 64, 37),
                                       (64, 36, 48, 20), (1920, 1080, 1280, 720)])
def test_nv12_i420_repack(vfhip, oracle, ifmt, ofmt, w, h, ow, oh, monkeypatch):
    """NV12 <-> I420 (the same matrix and siting): the (de)interleave with 16-byte accesses (k_repack_420_vec) where the frame allows, the
    byte-wise kernel elsewhere (width % 16 != 0, odd height), alone and as stage 1 of a scale — against the oracle and against each other"""
    rng = np.random.default_rng(w + 3 * oh)
    raw = rng.integers(0, 256, vfhip.plane_layout(ifmt, w, h)[1], dtype=np.uint8)
    want = oracle.convertscale(ifmt, w, h, raw, "bt709", "mpeg2", "bilinear", ofmt, ow, oh)
    got, kname = run(vfhip, ifmt, w, h, raw, "bt709", "mpeg2", "bilinear", ofmt, ow, oh)
    assert kname == "k_cs_staged_420", kname
    assert np.array_equal(meaningful(ofmt, ow, oh, got), meaningful(ofmt, ow, oh, want))
    monkeypatch.setenv("VFHIP_PLANE_SCALAR", "1")
    scalar, _ = run(vfhip, ifmt, w, h, raw, "bt709", "mpeg2", "bilinear", ofmt, ow, oh)
    monkeypatch.delenv("VFHIP_PLANE_SCALAR")
    assert np.array_equal(meaningful(ofmt, ow, oh, scalar), meaningful(ofmt, ow, oh, want))


@pytest.mark.gpu
@pytest.mark.parametrize("ifmt,ofmt", [("BGRA", "RGBA"), ("RGBA", "BGRA"), ("BGRA", "BGRA"), ("RGBA", "RGBA")])
@pytest.mark.parametrize("w,h", [(64, 36), (4, 1), (1920, 1080), (200, 113), (66, 5)])
@pytest.mark.parametrize("method", ["bilinear", "nearest", "bicubic"])
def test_rgb_same_size(vfhip, oracle, ifmt, ofmt, w, h, method, monkeypatch):
    """BGRA / RGBA -> BGRA / RGBA at the same size: a copy or the R <-> B swap (k_cs_rgb_same, 16 bytes per lane) for every method — videoscale
    passes through — against the oracle and against the kernels it replaces; a width that is not a multiple of 4 keeps those"""
    rng = np.random.default_rng(w + h)
    raw = rng.integers(0, 256, vfhip.plane_layout(ifmt, w, h)[1], dtype=np.uint8)
    want = oracle.convertscale(ifmt, w, h, raw, "bt709", "mpeg2", method, ofmt, w, h)
    got, kname = run(vfhip, ifmt, w, h, raw, "bt709", "mpeg2", method, ofmt, w, h)
    assert (kname == "k_cs_rgb_same") == (w % 4 == 0), kname
    assert np.array_equal(got, want)
    px = raw.reshape(h, w, 4)
    assert np.array_equal(got, px if ifmt == ofmt else px[..., [2, 1, 0, 3]])
    monkeypatch.setenv("VFHIP_NO_SAME", "1")
    old, kold = run(vfhip, ifmt, w, h, raw, "bt709", "mpeg2", method, ofmt, w, h)
    monkeypatch.delenv("VFHIP_NO_SAME")
    assert kold != "k_cs_rgb_same" and np.array_equal(old, want)


@pytest.mark.gpu
@pytest.mark.parametrize("ifmt,ofmt", [("BGRA", "BGRA"), ("RGBA", "BGRA"), ("BGRA", "RGBA")])
@pytest.mark.parametrize("w,h,ow,oh", [(192, 108, 128, 72), (200, 113, 67, 51), (1920, 1080, 1280, 720), (130, 70, 61, 69), (130, 70, 129, 31), (64, 36, 21, 9), (96, 54, 95, 53),
                                       (320, 180, 100, 179), (258, 258, 65, 65), (2, 2, 1, 1), (100, 64, 99, 200), (100, 64, 100, 30), (100, 64, 40, 64), (1000, 30, 64, 7), (3, 9, 2, 4)])
def test_rgb_taps_strip_kernel(vfhip, oracle, ifmt, ofmt, w, h, ow, oh, monkeypatch):
    """k_cs_rgb_taps_strip (RGB -> RGB bilinear with minification on an axis: one 8-byte window per source row, four output rows per lane) forced on
    whatever the launch size — partial strips and waves, the clamped last column, both pass orders, one axis only, the R <-> B swap — against
    the oracle and against k_cs_generic"""
    rng = np.random.default_rng(w * 17 + oh)
    raw = rng.integers(0, 256, vfhip.plane_layout(ifmt, w, h)[1], dtype=np.uint8)
    want = oracle.convertscale(ifmt, w, h, raw, "bt709", "mpeg2", "bilinear", ofmt, ow, oh)
    monkeypatch.setenv("VFHIP_TAPS_FILL", "0")
    got, kname = run(vfhip, ifmt, w, h, raw, "bt709", "mpeg2", "bilinear", ofmt, ow, oh)
    assert kname == "k_cs_generic", kname                      # (the name of the cell; strips are a launch-time choice)
    assert np.array_equal(got, want), "strips"
    monkeypatch.setenv("VFHIP_TAPS_ROWS", "1")
    one, _ = run(vfhip, ifmt, w, h, raw, "bt709", "mpeg2", "bilinear", ofmt, ow, oh)
    monkeypatch.delenv("VFHIP_TAPS_ROWS")
    monkeypatch.delenv("VFHIP_TAPS_FILL")
    assert np.array_equal(one, want), "one pixel per lane"


@pytest.mark.gpu
@pytest.mark.parametrize("ifmt,ofmt,w,h,ow,oh", [("NV12", "BGRA", 640, 360, 214, 160), ("I420", "RGBA", 640, 360, 426, 240), ("BGRA", "RGBA", 640, 360, 426, 240),
                                                  ("NV12", "I420", 640, 360, 640, 360), ("I420", "NV12", 640, 360, 640, 360), ("BGRA", "RGBA", 640, 360, 640, 360)])
def test_round2_late_kernels_on_the_batched_device_path(vfhip, oracle, ifmt, ofmt, w, h, ow, oh):
    """the strip kernels (NV12 / I420 / RGB down-scales), the 16-byte NV12 <-> I420 repack and the RGB same-size kernel on a batch of frames at a
    pitch, frame by frame against the oracle — large enough that the strips are the launch's own choice"""
    import torch
    n = 5
    rng = np.random.default_rng(w + ow)
    isz, osz = vfhip.plane_layout(ifmt, w, h)[1], vfhip.plane_layout(ofmt, ow, oh)[1]
    frames = [rng.integers(0, 256, isz, dtype=np.uint8) for _ in range(n)]
    want = [np.asarray(oracle.convertscale(ifmt, w, h, f, "bt709", "mpeg2", "bilinear", ofmt, ow, oh)).reshape(-1) for f in frames]
    cs = vfhip.ConvertScale(0)
    cs.configure(ifmt, w, h, ofmt, ow, oh, colorimetry="bt709", chroma_site="mpeg2")
    ip, op = (isz + 255) // 256 * 256, (osz + 255) // 256 * 256
    ring = np.zeros((n, ip), np.uint8)
    for k, f in enumerate(frames):
        ring[k, :isz] = f
    din, dout = torch.from_numpy(ring).cuda(), torch.zeros((n, op), dtype=torch.uint8, device="cuda")
    s = torch.cuda.Stream()
    torch.cuda.synchronize()
    cs.process_device(din.data_ptr(), dout.data_ptr(), stream=s.cuda_stream, n_frames=n, in_pitch=ip, out_pitch=op)
    s.synchronize()
    out = dout.cpu().numpy()
    for k in range(n):
        assert np.array_equal(meaningful(ofmt, ow, oh, out[k, :osz]), meaningful(ofmt, ow, oh, want[k])), k
    cs.close()
