"""GPU parity of the `metal`-numerics kernels (convertscale metal path, deinterlace, videofilter, compositor) through
the C ABI against the float oracle oracle/metalref.c.  Tolerance: +-1 LSB per byte (north_star), written below as TOL;
in practice both sides evaluate identical expressions with -ffp-contract=off, so almost every byte is identical and the
tests also bound the fraction of off-by-one bytes.  PARITY UNPINNED vs real Metal (see oracle/metalref.c header).

The video filter has two arithmetic paths (csrc/videofilter.hip): the default FAST one (hardware log / exp / rcp / sqrt like the reference's fast-math
MSL, uniforms folded on the host, fp16 LUT cells) and the EXACT one (VFHIP_VF_EXACT=1: the oracle's operation sequence).  Both are held to the oracle
here (fixture vf_mode).  A sharpened frame is checked stage by stage (vf_parity): an off-by-one byte of pass 1 — legitimate under +-1 — leaves the
unsharp mask multiplied by up to 1 + 2 |amount|, in the oracle's own arithmetic as much as in any other, so "+-1 against the one-shot oracle" is not
a property the reference's pipeline has; "+-1 per stage" is."""
import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.gpu
TOL = 1


def rnd(fmt, w, h, seed=0):
    return np.random.default_rng(seed).integers(0, 256, ol.raw_layout(fmt, w, h)[1], dtype=np.uint8)


def smooth(fmt, w, h, seed=0):
    """low-frequency content (so that bilinear / blur paths see realistic gradients) + a little noise"""
    rng = np.random.default_rng(seed)
    raw = np.zeros(ol.raw_layout(fmt, w, h)[1], np.uint8)
    pl, _ = ol.raw_layout(fmt, w, h)
    for i, (off, stride) in enumerate(pl):
        rows = h if i == 0 or fmt not in ("NV12", "I420") else (h + 1) // 2
        yy, xx = np.mgrid[0:rows, 0:stride]
        v = 128 + 90 * np.sin(xx / (7.0 + i)) * np.cos(yy / (5.0 + 2 * i)) + rng.integers(-6, 7, (rows, stride))
        raw[off:off + rows * stride] = np.clip(v, 0, 255).astype(np.uint8).reshape(-1)
    return raw


STATS = []


def close(got, want, what="", max_off_by_one=0.02):
    d = np.abs(got.astype(int) - want.astype(int))
    STATS.append((what, int(d.max()), float((d > 0).mean())))
    assert d.max() <= TOL, f"{what}: max diff {d.max()} at {np.argmax(d)} ({(d > TOL).sum()} bytes beyond tolerance)"
    assert (d > 0).mean() <= max_off_by_one, f"{what}: {(d > 0).mean():.4f} of bytes differ by 1"


@pytest.fixture(params=["fast", "exact"])
def vf_mode(request, monkeypatch):
    if request.param == "exact":
        monkeypatch.setenv("VFHIP_VF_EXACT", "1")
    else:
        monkeypatch.delenv("VFHIP_VF_EXACT", raising=False)
    return request.param


def vf_parity(vfhip, metalref, ifmt, ofmt, w, h, raw, kw, lut=None, m709=False, what="", max_off_by_one=0.05, got=None):
    """the GPU video filter against the oracle; returns the GPU frame.  Without sharpening: every byte within +-1, the differing fraction bounded.
    With sharpening, stage by stage: (1) pass 1 — everything but the sharpening, as BGRA — within +-1 of the oracle's, fraction bounded; (2) the
    GPU's own pass-1 frame through the ORACLE's blur + unsharp mask alone (and its output conversion) gives the GPU's sharpened frame, +-1 and
    almost everywhere equal; (3) against the one-shot oracle no byte is farther than 1 + ceil (2 |amount|), and bytes beyond +-1 are rare."""
    col = "bt709" if m709 else "bt601"
    prm = vfhip.filter_params(**kw)
    if got is None:
        vf = vfhip.VideoFilter(0)
        vf.configure(ifmt, w, h, ofmt, colorimetry=col)
        if lut is not None:
            vf.set_lut(lut)
        got = vf.process(raw, prm)
        vf.close()
    want = metalref.videofilter(ifmt, w, h, raw, ofmt, ol.mr_filter_params(prm), lut=lut, m709=m709)
    amount = kw.get("sharpness", 0.0)
    if not amount:
        close(got, want, what, max_off_by_one=max_off_by_one)
        return got
    kw1 = dict(kw, sharpness=0.0)
    prm1 = vfhip.filter_params(**kw1)
    vf = vfhip.VideoFilter(0)
    vf.configure(ifmt, w, h, "BGRA", colorimetry=col)
    if lut is not None:
        vf.set_lut(lut)
    p1 = vf.process(raw, prm1)
    vf.close()
    close(p1, metalref.videofilter(ifmt, w, h, raw, "BGRA", ol.mr_filter_params(prm1), lut=lut, m709=m709), what + " [pass 1]", max_off_by_one=max_off_by_one)
    stage2 = metalref.videofilter("BGRA", w, h, p1, ofmt, ol.mr_filter_params(vfhip.filter_params(sharpness=amount)), m709=m709)
    close(got, stage2, what + " [blur + unsharp mask of the GPU's pass 1]", max_off_by_one=0.005)
    d = np.abs(got.astype(int) - want.astype(int))
    STATS.append((what + " [one-shot]", int(d.max()), float((d > 0).mean())))
    assert d.max() <= 1 + int(np.ceil(2 * abs(amount))), f"{what}: {d.max()} from the one-shot oracle"
    assert (d > 1).mean() <= 0.002, f"{what}: {(d > 1).mean():.5f} of the bytes beyond +-1 of the one-shot oracle"
    return got


FORMATS6 = ["BGRA", "RGBA", "NV12", "I420", "UYVY", "YUY2"]


@pytest.mark.parametrize("ifmt", FORMATS6)
@pytest.mark.parametrize("ofmt", FORMATS6)
def test_convertscale_metal_matrix(vfhip, metalref, ifmt, ofmt):
    """the reference's full 6x6 format matrix (tests/test-convertscale.sh:62-99 shapes), metal numerics, with scaling"""
    for (w, h, ow, oh, method) in [(64, 36, 40, 30, "bilinear"), (33, 17, 66, 35, "nearest"), (48, 32, 48, 32, "bilinear"),
                                  (40, 30, 37, 21, "bilinear")]:      # odd output width: every byte of a packed row is defined
        raw = smooth(ifmt, w, h, 3)
        cs = vfhip.ConvertScale(0)
        cs.configure(ifmt, w, h, ofmt, ow, oh, method=method, numerics="metal", colorimetry="bt709")
        assert cs.kernel_name == "k_cs_metal"
        got = cs.process(raw)
        cs.close()
        want = metalref.convertscale(ifmt, w, h, raw, ofmt, ow, oh, linear=(method == "bilinear"), m709_in=True, m709_out=True)
        close(got, want, f"{ifmt}->{ofmt} {w}x{h}->{ow}x{oh} {method}")


def test_convertscale_gst_exact_falls_back_to_metal_for_unpinned_cells(vfhip, metalref):
    raw = smooth("UYVY", 64, 32, 1)
    cs = vfhip.ConvertScale(0)
    cs.configure("UYVY", 64, 32, "NV12", 30, 30, add_borders=True, numerics="gst-exact")     # borders whose rectangle (30 x 15) is not on chroma-sample boundaries: not pinned
    assert cs.kernel_name == "k_cs_metal"
    assert cs.numerics_in_effect == "metal"                      # ... and says so: never a silent substitution
    close(cs.process(raw), metalref.convertscale("UYVY", 64, 32, raw, "NV12", 30, 30, add_borders=True), "UYVY->NV12 borders")
    # gst-exact-strict refuses the cell instead (VFHIP_ERR_UNSUPPORTED) and leaves the handle unconfigured
    with pytest.raises(vfhip.VfHipError) as e:
        cs.configure("UYVY", 64, 32, "NV12", 30, 30, add_borders=True, numerics="gst-exact-strict")
    assert e.value.code == -2
    # a pinned cell configures under both spellings and reports gst-exact
    for num in ("gst-exact", "gst-exact-strict"):
        cs.configure("NV12", 64, 32, "BGRA", 32, 16, numerics=num, colorimetry="bt709", chroma_site="mpeg2")
        assert cs.numerics_in_effect == "gst-exact" and cs.kernel_name == "k_cs_nv12_half"
    cs.configure("NV12", 64, 32, "BGRA", 32, 16, numerics="metal")
    assert cs.numerics_in_effect == "metal"
    # YUV -> YUV with a matrix change and NV12 <-> I420 with a siting change ARE pinned (round 2: k_yuv_to_yuv): exact under both spellings
    for kw in (dict(colorimetry="bt709", out_colorimetry="bt601"), dict(chroma_site="mpeg2", out_chroma_site="jpeg")):
        for num in ("gst-exact", "gst-exact-strict"):
            cs.configure("NV12", 64, 32, "I420", 64, 32, numerics=num, **kw)
            assert cs.numerics_in_effect == "gst-exact" and cs.kernel_name == "k_cs_staged_420", kw
    cs.close()


@pytest.mark.parametrize("w,h,ow,oh", [(64, 16, 32, 32), (16, 64, 32, 32), (40, 30, 64, 64), (64, 36, 64, 36)])
def test_convertscale_letterbox(vfhip, metalref, w, h, ow, oh):
    raw = smooth("BGRA", w, h, 2)
    cs = vfhip.ConvertScale(0)
    cs.configure("BGRA", w, h, "BGRA", ow, oh, add_borders=True, border_color=0x80FF2010, numerics="metal")
    close(cs.process(raw), metalref.convertscale("BGRA", w, h, raw, "BGRA", ow, oh, add_borders=True, border=0x80FF2010), "letterbox")
    cs.close()


@pytest.mark.parametrize("fmt", ["BGRA", "RGBA", "NV12", "I420"])
@pytest.mark.parametrize("method", ["bob", "weave", "linear", "greedyh"])
def test_deinterlace_methods(vfhip, metalref, fmt, method):
    """4 methods x 4 formats (reference tests/test-deinterlace.sh), 3 frames so that the history path is exercised;
    odd and even sizes; both field orders."""
    for (w, h, tff) in [(64, 36, True), (33, 19, False)]:
        frames = [smooth(fmt, w, h, 10 + k) for k in range(3)]
        frames[2] = frames[1].copy()                       # a static frame: greedyh must weave
        d = vfhip.Deinterlace(0)
        d.configure(fmt, w, h, colorimetry="bt709")
        prev = None
        for k, f in enumerate(frames):
            got = d.process(f, method=method, tff=tff, threshold=0.08)
            want = metalref.deinterlace(fmt, w, h, f, prev, vfhip.DEINTERLACE_METHODS[method], tff=tff, threshold=0.08, m709=True)
            close(got, want, f"{fmt} {method} frame {k} {w}x{h}")
            prev = f
        d.reset()                                          # history dropped: weave / greedyh fall back to bob
        got = d.process(frames[0], method=method, tff=tff)
        want = metalref.deinterlace(fmt, w, h, frames[0], None, vfhip.DEINTERLACE_METHODS[method], tff=tff, m709=True)
        close(got, want, f"{fmt} {method} after reset")
        d.close()


def test_deinterlace_device_path_history(vfhip, metalref):
    import torch
    fmt, w, h = "NV12", 128, 72
    size = ol.raw_layout(fmt, w, h)[1]
    frames = [smooth(fmt, w, h, 20 + k) for k in range(3)]
    d = vfhip.Deinterlace(0)
    d.configure(fmt, w, h)
    s = torch.cuda.Stream()
    prev = None
    for f in frames:
        din = torch.from_numpy(f).cuda()
        dout = torch.zeros(size, dtype=torch.uint8, device="cuda")
        s.wait_stream(torch.cuda.current_stream())
        d.process_device(din.data_ptr(), dout.data_ptr(), method="greedyh", tff=True, threshold=0.05, stream=s.cuda_stream)
        s.synchronize()
        close(dout.cpu().numpy(), metalref.deinterlace(fmt, w, h, f, prev, 3, threshold=0.05), "device greedyh")
        prev = f
    d.close()


@pytest.mark.parametrize("n,per_row", [(8, 8), (16, 4), (4, 2)])
def test_videofilter_png_lut_file(vfhip, metalref, tmp_path, n, per_row):
    """PNG LUT layout of the reference (parse_png_lut, metalvideofilterrenderer.m:166-305): N x N slices, width / N per row"""
    import png_util
    rng = np.random.default_rng(n)
    lut8 = rng.integers(0, 256, (n, n, n, 3), dtype=np.uint8)              # [b][g][r]
    rows = (n + per_row - 1) // per_row
    img = np.zeros((rows * n, per_row * n, 3), np.uint8)
    for b in range(n):
        img[(b // per_row) * n:(b // per_row + 1) * n, (b % per_row) * n:(b % per_row + 1) * n] = lut8[b]
    path = tmp_path / "lut.png"
    png_util.write_png(path, img, 2, 8, filters=[1, 4])
    w, h = 48, 24
    raw = smooth("BGRA", w, h, 9)
    vf = vfhip.VideoFilter(0)
    vf.configure("BGRA", w, h)
    vf.load_lut(str(path))
    assert vf.lut_size == n
    lut = np.ones((n, n, n, 4), np.float32)
    lut[..., :3] = lut8.astype(np.float32) / np.float32(255.0)
    prm = vfhip.filter_params(contrast=1.1)
    close(vf.process(raw, prm), metalref.videofilter("BGRA", w, h, raw, "BGRA", ol.mr_filter_params(prm), lut=lut), "png lut", max_off_by_one=0.05)
    png_util.write_png(tmp_path / "odd.png", np.zeros((5, 7, 3), np.uint8), 2)          # 35 pixels: no cube
    with pytest.raises(vfhip.VfHipError):
        vf.load_lut(str(tmp_path / "odd.png"))
    assert vf.lut_size == n
    vf.close()


SINGLE = [dict(brightness=0.3), dict(brightness=-0.4), dict(contrast=1.7), dict(contrast=0.3), dict(saturation=0.0), dict(saturation=1.8),
          dict(hue=1.0), dict(hue=-2.5), dict(gamma=2.2), dict(gamma=0.45), dict(sepia=0.8), dict(invert=True), dict(vignette=0.9),
          dict(chroma_key=(0.0, 1.0, 0.0), tolerance=0.3, smoothness=0.1), dict(chroma_key=(0.5, 0.5, 0.5), tolerance=0.2, smoothness=0.0),
          dict(sharpness=0.8), dict(sharpness=-0.6)]
# the reference's "all colour adjustments" set (tests/test-videofilter.sh:198-201) + invert + chroma key (:183-186)
ALL15 = dict(brightness=0.1, contrast=1.2, saturation=0.8, hue=0.3 * np.pi, gamma=1.5, sharpness=0.5, sepia=0.2, vignette=0.3,
             invert=True, chroma_key=(0.0, 1.0, 0.0), tolerance=0.3, smoothness=0.1)


@pytest.mark.parametrize("kw", SINGLE, ids=[",".join(k) for k in SINGLE])
def test_videofilter_single_properties(vfhip, metalref, kw, vf_mode):
    # (brightness .3 puts EVERY byte on a tie, x + 76.5: where two float evaluation orders round such a value is noise — 3 % of them differ)
    w, h = 96, 40
    vf_parity(vfhip, metalref, "BGRA", "BGRA", w, h, smooth("BGRA", w, h, 5), kw, what=f"{vf_mode} {kw}", max_off_by_one=0.05)
    vf_parity(vfhip, metalref, "RGBA", "RGBA", 100, 52, rnd("RGBA", 100, 52, 15), kw, what=f"{vf_mode} random frame {kw}", max_off_by_one=0.05)


@pytest.mark.parametrize("ifmt,ofmt", [("BGRA", "BGRA"), ("RGBA", "BGRA"), ("NV12", "NV12"), ("I420", "I420"), ("NV12", "BGRA"), ("BGRA", "I420")])
@pytest.mark.parametrize("w,h", [(96, 40), (71, 37)])
def test_videofilter_all_effects_and_formats(vfhip, metalref, ifmt, ofmt, w, h, vf_mode):
    raw = smooth(ifmt, w, h, 6)
    n = 9
    g = np.linspace(0, 1, n, dtype=np.float32)
    lut = np.ones((n, n, n, 4), np.float32)
    lut[..., 0] = g[None, None, :] ** 1.1
    lut[..., 1] = g[None, :, None] * 0.9
    lut[..., 2] = 1.0 - g[:, None, None]
    vf = vfhip.VideoFilter(0)
    vf.configure(ifmt, w, h, ofmt, colorimetry="bt709")
    vf.set_lut(lut)
    assert vf.lut_size == n
    prm = vfhip.filter_params(**ALL15)
    got = vf.process(raw, prm)
    vf.clear_lut()
    assert vf.lut_size == 0
    got_nolut = vf.process(raw, prm)
    vf.close()
    # (a LUT that runs against the identity — blue inverted — is the fp16 residual table's worst case: the residual is as large as the value)
    vf_parity(vfhip, metalref, ifmt, ofmt, w, h, raw, ALL15, lut=lut, m709=True, what=f"{vf_mode} all15 {ifmt}->{ofmt}", max_off_by_one=0.08, got=got)
    vf_parity(vfhip, metalref, ifmt, ofmt, w, h, raw, ALL15, m709=True, what=f"{vf_mode} all15 no lut {ifmt}->{ofmt}", max_off_by_one=0.08, got=got_nolut)
    for kw in (dict(ALL15, sharpness=0.0), dict(ALL15, sharpness=-0.7), dict(ALL15, sharpness=1.0)):
        vf_parity(vfhip, metalref, ifmt, ofmt, w, h, raw, kw, lut=lut, m709=True, what=f"{vf_mode} all15 sharpness {kw['sharpness']} {ifmt}->{ofmt}", max_off_by_one=0.08)


def test_videofilter_noise_statistics(vfhip, metalref):
    """float32 fract chains amplify ULP differences: compare statistically"""
    w, h = 128, 64
    raw = np.full(ol.raw_layout("BGRA", w, h)[1], 128, np.uint8)
    vf = vfhip.VideoFilter(0)
    vf.configure("BGRA", w, h)
    outs = []
    for frame in (0, 1):
        prm = vfhip.filter_params(noise=0.5, frame_index=frame)
        got = vf.process(raw, prm).reshape(h, w, 4)[..., :3].astype(float)
        want = metalref.videofilter("BGRA", w, h, raw, "BGRA", ol.mr_filter_params(prm)).reshape(h, w, 4)[..., :3].astype(float)
        assert abs(got.mean() - want.mean()) < 1.0 and abs(got.std() - want.std()) < 1.0
        assert (np.abs(got - want) <= 1).mean() > 0.98          # the hash is well conditioned almost everywhere
        assert got.std() > 10                                    # noise=.5 -> +-0.125 uniform -> sigma ~ 18 LSB
        outs.append(got)
    assert np.corrcoef(outs[0].ravel(), outs[1].ravel())[0, 1] < 0.2   # per-frame decorrelation
    vf.close()


def test_videofilter_cube_lut_file(vfhip, metalref, tmp_path):
    n = 5
    path = tmp_path / "t.cube"
    rows = ["# comment", "TITLE \"t\"", f"LUT_3D_SIZE {n}", "DOMAIN_MIN 0 0 0", "DOMAIN_MAX 1 1 1"]
    lut = np.ones((n, n, n, 4), np.float32)
    for b in range(n):
        for g in range(n):
            for r in range(n):
                v = (r / (n - 1) * 0.5, g / (n - 1), 1 - b / (n - 1))
                lut[b, g, r, :3] = v
                rows.append("%.6f %.6f %.6f" % v)
    path.write_text("\n".join(rows) + "\n")
    w, h = 48, 24
    raw = smooth("RGBA", w, h, 8)
    vf = vfhip.VideoFilter(0)
    vf.configure("RGBA", w, h)
    vf.load_lut(str(path))
    assert vf.lut_size == n
    prm = vfhip.filter_params()
    lut = np.float32(np.round(lut, 6))
    close(vf.process(raw, prm), metalref.videofilter("RGBA", w, h, raw, "RGBA", ol.mr_filter_params(prm), lut=lut), "cube", max_off_by_one=0.05)
    with pytest.raises(vfhip.VfHipError) as e:
        vf.load_lut(str(tmp_path / "missing.cube"))
    assert e.value.code == -7
    with pytest.raises(vfhip.VfHipError) as e:
        vf.load_lut(str(tmp_path / "lut.txt"))
    assert e.value.code == -2
    with pytest.raises(vfhip.VfHipError):
        vf.load_lut(str(tmp_path / "missing.png"))
    (tmp_path / "bad.cube").write_text("LUT_3D_SIZE 3\n0 0 0\n")
    with pytest.raises(vfhip.VfHipError):
        vf.load_lut(str(tmp_path / "bad.cube"))
    assert vf.lut_size == n                                       # a failed load leaves the old LUT in place
    vf.close()


def test_videofilter_1080p_c3_config(vfhip, metalref, vf_mode):
    """BASELINE config 2 shape: BGRA 1920x1080, all 15 properties + a 33^3 LUT, whole frame against the oracle (vignette / texcoord depend on the
    full frame size: about 2 s on the CPU per oracle pass), on a smooth frame and on uniform random bytes (what the bench feeds it); determinism"""
    w, h = 1920, 1080
    n = 33
    g = np.linspace(0, 1, n, dtype=np.float32)
    lut = np.ones((n, n, n, 4), np.float32)
    lut[..., 0], lut[..., 1], lut[..., 2] = g[None, None, :] ** 1.05, g[None, :, None], g[:, None, None] ** 0.95
    kw = dict(ALL15, noise=0.0)
    for name, raw in (("smooth", smooth("BGRA", w, h, 9)), ("random", rnd("BGRA", w, h, 19))):
        prm = vfhip.filter_params(**kw)
        vf = vfhip.VideoFilter(0)
        vf.configure("BGRA", w, h)
        vf.set_lut(lut)
        a = vf.process(raw, prm)
        b = vf.process(raw, prm)
        vf.close()
        assert np.array_equal(a, b)
        vf_parity(vfhip, metalref, "BGRA", "BGRA", w, h, raw, kw, lut=lut, what=f"{vf_mode} C3 1080p {name}", max_off_by_one=0.02, got=a)


def test_videofilter_fast_path_against_exact_path_1080p(vfhip, monkeypatch):
    """the two arithmetic paths of the kernels against each other on a full random 1080p frame, colour stages + LUT (no sharpening): +-1 and rare;
    and the fast path on the fp32 table (VFHIP_VF_LUT32) against the fp16 one: what fp16 cells cost in differing bytes, here with a grading LUT"""
    w, h = 1920, 1080
    raw = rnd("BGRA", w, h, 29)
    n = 17
    g = np.linspace(0, 1, n, dtype=np.float32)
    lut = np.ones((n, n, n, 4), np.float32)
    lut[..., 0], lut[..., 1], lut[..., 2] = g[None, None, :] ** 0.9, g[None, :, None] * 0.95 + 0.02, g[:, None, None] ** 1.2
    prm = vfhip.filter_params(**dict(ALL15, sharpness=0.0, noise=0.05))
    vf = vfhip.VideoFilter(0)
    vf.configure("BGRA", w, h)
    vf.set_lut(lut)
    monkeypatch.delenv("VFHIP_VF_EXACT", raising=False)
    fast = vf.process(raw, prm)
    monkeypatch.setenv("VFHIP_VF_LUT32", "1")
    fast32 = vf.process(raw, prm)
    monkeypatch.delenv("VFHIP_VF_LUT32")
    monkeypatch.setenv("VFHIP_VF_EXACT", "1")
    exact = vf.process(raw, prm)
    vf.close()
    close(fast32, exact, "fast colour stages vs exact, fp32 table", max_off_by_one=0.002)
    close(fast, exact, "fast path vs exact path", max_off_by_one=0.01)


def pads_case(w, h, seed=0):
    a = smooth("BGRA", 64, 48, seed)
    a.reshape(-1, 4)[:, 3] = np.random.default_rng(seed).integers(0, 256, 64 * 48)
    b = smooth("NV12", 40, 30, seed + 1)
    c = smooth("I420", 33, 21, seed + 2)
    d = smooth("RGBA", 16, 16, seed + 3)
    return [("BGRA", 64, 48, a, 4, 6, 64, 48, 0.9, 1),            # unscaled
            ("NV12", 40, 30, b, 30, 10, 60, 45, 0.7, 1, True),    # upscaled, bt709
            ("I420", 33, 21, c, -10, 40, 50, 30, 1.0, 2),         # partly off-screen, ADD
            ("RGBA", 16, 16, d, w - 20, h - 12, 32, 24, 0.5, 0)]  # hangs over the corner, SOURCE


def to_vf(vfhip, pads):
    inv = {0: "source", 1: "over", 2: "add"}
    return [(p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7], p[8], inv[p[9]], "bt709" if len(p) > 10 and p[10] else "bt601") for p in pads]


@pytest.mark.parametrize("bg", ["checker", "black", "white", "transparent"])
@pytest.mark.parametrize("ofmt", ["BGRA", "RGBA", "NV12", "I420"])
def test_compositor_backgrounds_formats(vfhip, metalref, bg, ofmt):
    w, h = 100, 75
    pads = pads_case(w, h, 4)
    comp = vfhip.Compositor(0)
    comp.configure(ofmt, w, h, colorimetry="bt709")
    got = comp.composite(to_vf(vfhip, pads), background=bg)
    want = metalref.compositor(ofmt, w, h, pads, vfhip.BACKGROUNDS[bg], m709_out=True)
    close(got, want, f"compositor {bg} {ofmt}")
    comp.close()


def test_compositor_no_pads_and_many_pads(vfhip, metalref):
    w, h = 64, 40
    comp = vfhip.Compositor(0)
    comp.configure("BGRA", w, h)
    assert np.array_equal(comp.composite([], background="checker"), metalref.compositor("BGRA", w, h, [], 0))
    rng = np.random.default_rng(1)
    pads = []
    for k in range(37):                                        # > 16 layers: chained passes through the RGBA8 scratch
        pw, ph = int(rng.integers(4, 30)), int(rng.integers(4, 30))
        raw = rnd("BGRA", pw, ph, 100 + k)
        pads.append(("BGRA", pw, ph, raw, int(rng.integers(-8, w)), int(rng.integers(-8, h)), int(rng.integers(4, 40)), int(rng.integers(4, 40)),
                     float(rng.uniform(0.2, 1.0)), int(rng.integers(0, 3))))
    got = comp.composite(to_vf(vfhip, pads), background="black")
    close(got, metalref.compositor("BGRA", w, h, pads, 1), "37 pads", max_off_by_one=0.05)
    comp.close()


def test_compositor_c4_config(vfhip, metalref):
    """BASELINE config 3: 4 x BGRA 1080p quadrants (alpha .9, over, z 0-3) + NV12 720p centred (alpha .7, z 4) -> 2160p,
    background black.  Whole frame vs oracle."""
    w, h = 3840, 2160
    quads = [smooth("BGRA", 1920, 1080, 30 + k) for k in range(4)]
    nv = smooth("NV12", 1280, 720, 40)
    pads = [("BGRA", 1920, 1080, quads[k], (k % 2) * 1920, (k // 2) * 1080, 1920, 1080, 0.9, 1) for k in range(4)]
    pads.append(("NV12", 1280, 720, nv, (w - 1280) // 2, (h - 720) // 2, 1280, 720, 0.7, 1, True))
    comp = vfhip.Compositor(0)
    comp.configure("BGRA", w, h)
    got = comp.composite(to_vf(vfhip, pads), background="black")
    close(got, metalref.compositor("BGRA", w, h, pads, 1), "C4")
    comp.close()


@pytest.mark.parametrize("method", ["none", "clockwise", "rotate-180", "counterclockwise", "horizontal-flip", "vertical-flip",
                                    "upper-left-diagonal", "upper-right-diagonal"])
@pytest.mark.parametrize("fmt", ["BGRA", "NV12", "I420", "RGBA"])
def test_transform_methods(vfhip, metalref, method, fmt):
    """8 methods x 4 formats x crop (reference tests/test-transform.sh shapes), incl. odd sizes"""
    for (w, h, crop) in [(64, 48, (0, 0, 0, 0)), (61, 35, (3, 5, 7, 2)), (48, 48, (0, 8, 0, 0))]:
        raw = smooth(fmt, w, h, 50)
        t = vfhip.Transform(0)
        t.configure(fmt, w, h, colorimetry="bt709")
        got = t.process(raw, method=method, crop=crop)
        want = metalref.transform(fmt, w, h, raw, fmt, vfhip.TRANSFORM_METHODS[method], crop, m709=True)
        close(got, want, f"transform {method} {fmt} {w}x{h} crop {crop}")
        t.close()


def test_transform_exact_flips(vfhip):
    """square RGBA frame: the flips / rotations are pure pixel permutations (texel centres map onto texel centres)"""
    w = h = 32
    raw = rnd("RGBA", w, h, 51)
    img = raw.reshape(h, w, 4)
    t = vfhip.Transform(0)
    t.configure("RGBA", w, h)
    exp = {"none": img, "horizontal-flip": img[:, ::-1], "vertical-flip": img[::-1], "rotate-180": img[::-1, ::-1],
           "upper-left-diagonal": img.transpose(1, 0, 2), "clockwise": img.transpose(1, 0, 2)[:, ::-1],
           "counterclockwise": img.transpose(1, 0, 2)[::-1]}
    for m, e in exp.items():
        got = t.process(raw, method=m).reshape(h, w, 4)
        assert np.abs(got.astype(int) - e.astype(int)).max() <= 1, m
    t.close()


def test_metal_element_errors(vfhip):
    import ctypes as C
    d = vfhip.Deinterlace(0)
    fi, fo, prm = vfhip.Frame(), vfhip.Frame(), vfhip.DeinterlaceParams(0, 1, 0.1, 0)
    assert vfhip.lib.vfhip_deinterlace_process(d.h, C.byref(fi), C.byref(fo), C.byref(prm)) == -3
    with pytest.raises(vfhip.VfHipError):
        d.configure("UYVY", 16, 16)
    d.close()
    vf = vfhip.VideoFilter(0)
    with pytest.raises(vfhip.VfHipError):
        vf.in_fmt = vf.out_fmt = "BGRA"
        a, b = vfhip.make_info("BGRA", 16, 16), vfhip.make_info("BGRA", 32, 16)
        vfhip.check(vfhip.lib.vfhip_videofilter_configure(vf.h, C.byref(a), C.byref(b)))
    vf.close()
    c = vfhip.Compositor(0)
    assert vfhip.lib.vfhip_compositor_composite(c.h, None, 0, 0, C.byref(fo)) == -3
    c.close()


def test_zz_report_exactness():
    """not a check: prints how close to bit-exact the float kernels are against the oracle (pytest -s / -rP shows it)"""
    worst = sorted(STATS, key=lambda t: -t[2])[:8]
    print("comparisons:", len(STATS), "bit-exact:", sum(1 for t in STATS if t[1] == 0), "worst off-by-one fractions:", worst)


# ---- batched device entry points (frame k at base + k * pitch): every frame of a batch == the oracle on that frame ------
def _ring(frames, pitch):
    import torch
    buf = np.zeros((len(frames), pitch), np.uint8)
    for k, f in enumerate(frames):
        buf[k, :f.size] = f
    return torch.from_numpy(buf).cuda()


@pytest.mark.parametrize("fmt", ["BGRA", "NV12", "I420"])
def test_deinterlace_batch_is_a_stream(vfhip, metalref, fmt):
    """a batch is n consecutive frames of one stream: history of frame k = frame k-1, and the handle's history carries
    across batches (two batches of 3 == six sequential frames)"""
    import torch
    w, h, n = 130, 74, 6
    size = ol.raw_layout(fmt, w, h)[1]
    pitch = (size + 255) // 256 * 256
    frames = [smooth(fmt, w, h, 40 + k) for k in range(n)]
    d = vfhip.Deinterlace(0)
    d.configure(fmt, w, h)
    s = torch.cuda.Stream()
    din, dout = _ring(frames, pitch), torch.zeros((n, pitch), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    for b in range(2):
        d.process_device(din[3 * b].data_ptr(), dout[3 * b].data_ptr(), method="greedyh", tff=False, threshold=0.05, stream=s.cuda_stream,
                         n_frames=3, in_pitch=pitch, out_pitch=pitch)
    s.synchronize()
    out = dout.cpu().numpy()
    for k in range(n):
        want = metalref.deinterlace(fmt, w, h, frames[k], frames[k - 1] if k else None, 3, tff=False, threshold=0.05)
        close(out[k, :size], want, f"batch deinterlace {fmt} frame {k}")
    d.close()


@pytest.mark.parametrize("sharp", [0.0, 0.6])
def test_videofilter_batch(vfhip, metalref, sharp, vf_mode):
    import torch
    w, h, n = 100, 44, 4
    size = 4 * w * h
    pitch = (size + 255) // 256 * 256
    frames = [smooth("RGBA", w, h, 60 + k) for k in range(n)]
    vf = vfhip.VideoFilter(0)
    vf.configure("RGBA", w, h, "NV12")
    osize = ol.raw_layout("NV12", w, h)[1]
    opitch = (osize + 255) // 256 * 256
    kw = dict(brightness=0.05, contrast=1.1, gamma=1.3, noise=0.2, vignette=0.4, sharpness=sharp, frame_index=7)
    prm = vfhip.filter_params(**kw)
    s = torch.cuda.Stream()
    din, dout = _ring(frames, pitch), torch.zeros((n, opitch), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    vf.process_device(din.data_ptr(), dout.data_ptr(), prm, stream=s.cuda_stream, n_frames=n, in_pitch=pitch, out_pitch=opitch)
    s.synchronize()
    out = dout.cpu().numpy()
    for k in range(n):
        pk = ol.mr_filter_params(vfhip.filter_params(**dict(kw, frame_index=7 + k)))      # frame_index advances per frame
        vf_parity(vfhip, metalref, "RGBA", "NV12", w, h, frames[k], dict(kw, frame_index=7 + k), what=f"{vf_mode} batch videofilter frame {k}", max_off_by_one=0.05, got=out[k, :osize])
        assert pk.frame_index == 7 + k
    vf.close()


def test_compositor_batch(vfhip, metalref):
    import torch
    ow, oh, n = 96, 64, 3
    a = [smooth("BGRA", 64, 48, 70 + k) for k in range(n)]
    b = [smooth("NV12", 40, 30, 80 + k) for k in range(n)]
    logo = smooth("RGBA", 16, 16, 90)                          # pitch 0: the same frame in every output
    pa, pb = (a[0].size + 255) // 256 * 256, (b[0].size + 255) // 256 * 256
    da, db, dl = _ring(a, pa), _ring(b, pb), torch.from_numpy(logo).cuda()
    dout = torch.zeros((n, ow * oh * 4), dtype=torch.uint8, device="cuda")
    comp = vfhip.Compositor(0)
    comp.configure("BGRA", ow, oh)
    pads = [comp.pad("BGRA", 64, 48, da.data_ptr(), 0, 0, 64, 48, 0.9, "over"),
            comp.pad("NV12", 40, 30, db.data_ptr(), 30, 20, 60, 40, 0.7, "add"),
            comp.pad("RGBA", 16, 16, dl.data_ptr(), 70, 4, 16, 16, 1.0, "over")]
    s = torch.cuda.Stream()
    torch.cuda.synchronize()
    comp.composite_device(pads, dout.data_ptr(), background="checker", stream=s.cuda_stream, n_frames=n, pad_pitches=[pa, pb, 0], out_pitch=ow * oh * 4)
    s.synchronize()
    out = dout.cpu().numpy()
    for k in range(n):
        want = metalref.compositor("BGRA", ow, oh, [("BGRA", 64, 48, a[k], 0, 0, 64, 48, 0.9, 1), ("NV12", 40, 30, b[k], 30, 20, 60, 40, 0.7, 2),
                                                    ("RGBA", 16, 16, logo, 70, 4, 16, 16, 1.0, 1)], 0)
        close(out[k], want, f"batch compositor frame {k}")
    comp.close()


def test_transform_batch(vfhip, metalref):
    import torch
    w, h, n = 66, 38, 3
    size = ol.raw_layout("I420", w, h)[1]
    pitch = (size + 255) // 256 * 256
    frames = [smooth("I420", w, h, 95 + k) for k in range(n)]
    t = vfhip.Transform(0)
    t.configure("I420", w, h)
    din, dout = _ring(frames, pitch), torch.zeros((n, pitch), dtype=torch.uint8, device="cuda")
    s = torch.cuda.Stream()
    torch.cuda.synchronize()
    t.process_device(din.data_ptr(), dout.data_ptr(), method="rotate-180", crop=(2, 4, 6, 0), stream=s.cuda_stream, n_frames=n, in_pitch=pitch, out_pitch=pitch)
    s.synchronize()
    out = dout.cpu().numpy()
    for k in range(n):
        close(out[k, :size], metalref.transform("I420", w, h, frames[k], "I420", 2, crop=(2, 4, 6, 0)), f"batch transform frame {k}")
    t.close()


# ---- overlay ---------------------------------------------------------------------------------------------------------
def _logo(w, h, seed=0):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.zeros((h, w, 4), np.uint8)
    img[..., 0] = (xx * 255 // max(w - 1, 1)); img[..., 1] = (yy * 255 // max(h - 1, 1)); img[..., 2] = rng.integers(0, 256, (h, w))
    img[..., 3] = np.clip(255 - 6 * np.hypot(xx - w / 2, yy - h / 2), 0, 255)          # soft round alpha
    return img


@pytest.mark.parametrize("ifmt,ofmt", [("BGRA", "BGRA"), ("RGBA", "NV12"), ("NV12", "NV12"), ("I420", "BGRA"), ("NV12", "I420")])
def test_overlay_formats_and_placement(vfhip, metalref, ifmt, ofmt):
    w, h = 96, 54
    raw = smooth(ifmt, w, h, 11)
    img = _logo(24, 16)
    ov = vfhip.Overlay(0)
    ov.configure(ifmt, w, h, ofmt, colorimetry="bt709")
    ov.set_image(img)
    assert ov.image_size == (24, 16)
    for kw in (dict(x=10, y=7, alpha=0.8), dict(x=-5, y=40, width=50, height=30, alpha=1.0), dict(x=80, y=0, width=0, height=9, alpha=0.35),
               dict(x=30.5, y=12.25, width=33.3, height=20.7, alpha=0.6), dict(x=200, y=200, alpha=1.0)):
        got = ov.process(raw, **kw)
        want = metalref.overlay(ifmt, w, h, raw, ofmt, img, m709=True, **kw)
        close(got, want, f"overlay {ifmt}->{ofmt} {kw}", max_off_by_one=0.03)
    ov.clear_image()
    assert ov.image_size is None
    close(ov.process(raw), metalref.overlay(ifmt, w, h, raw, ofmt, None, m709=True), "overlay without image")
    ov.close()


def test_overlay_png_file_and_batch(vfhip, metalref, tmp_path):
    import png_util
    import torch
    w, h, n = 80, 48, 3
    img = _logo(20, 20, 3)
    path = tmp_path / "logo.png"
    png_util.write_png(path, img, 6, 8, filters=[4, 2])
    ov = vfhip.Overlay(0)
    ov.configure("BGRA", w, h)
    ov.load_image(str(path))
    pre = img.copy()
    pre[..., :3] = (img[..., :3].astype(np.uint32) * img[..., 3:4] + 127) // 255            # what the loader premultiplies to
    frames = [smooth("BGRA", w, h, 20 + k) for k in range(n)]
    close(ov.process(frames[0], x=30, y=10, alpha=0.9), metalref.overlay("BGRA", w, h, frames[0], "BGRA", pre, x=30, y=10, alpha=0.9), "png overlay")
    pitch = (4 * w * h + 255) // 256 * 256
    din, dout = _ring(frames, pitch), torch.zeros((n, pitch), dtype=torch.uint8, device="cuda")
    s = torch.cuda.Stream()
    torch.cuda.synchronize()
    ov.process_device(din.data_ptr(), dout.data_ptr(), x=5, y=5, width=40, height=30, alpha=0.5, stream=s.cuda_stream, n_frames=n, in_pitch=pitch, out_pitch=pitch)
    s.synchronize()
    out = dout.cpu().numpy()
    for k in range(n):
        close(out[k, :4 * w * h], metalref.overlay("BGRA", w, h, frames[k], "BGRA", pre, x=5, y=5, width=40, height=30, alpha=0.5), f"batch overlay {k}")
    with pytest.raises(vfhip.VfHipError) as e:
        ov.load_image(str(tmp_path / "missing.jpg"))             # a failed load keeps the current image
    assert e.value.code == -1 and ov.image_size == (20, 20)
    bad = tmp_path / "text.png"
    bad.write_bytes(b"neither a PNG nor a JPEG file" * 4)
    with pytest.raises(vfhip.VfHipError) as e:
        ov.load_image(str(bad))
    assert e.value.code == -2 and ov.image_size == (20, 20)
    ov.load_image("")                                       # empty path clears, like the reference
    assert ov.image_size is None
    ov.close()


def _mixed_pads(w, h, n, seed):
    """pads of every kind in random z-order: unscaled RGBA / BGRA at any position (the lean kernel's), unscaled NV12 / I420 on and off
    the chroma grid, scaled ones; some partly or wholly outside the frame"""
    rng = np.random.default_rng(seed)
    pads = []
    for k in range(n):
        fmt = ["BGRA", "RGBA", "BGRA", "NV12", "I420"][int(rng.integers(0, 5))]
        pw, ph = int(rng.integers(3, 150)), int(rng.integers(3, 90))
        if fmt in ("NV12", "I420") and rng.integers(0, 2):
            pw, ph = pw & ~1 | 8, ph & ~1 | 8
        raw = smooth(fmt, pw, ph, 1000 * seed + k)
        if fmt in ("BGRA", "RGBA"):
            raw.reshape(-1, 4)[:, 3] = rng.integers(0, 256, pw * ph)
        scaled = rng.integers(0, 4) == 0
        dw, dh = (int(rng.integers(4, 200)), int(rng.integers(4, 120))) if scaled else (pw, ph)
        x, y = int(rng.integers(-40, w)), int(rng.integers(-30, h))
        if not scaled and rng.integers(0, 2):
            x, y = x & ~3, y & ~3                               # half of the unscaled pads on the lane grid
        pads.append((fmt, pw, ph, raw, x, y, dw, dh, float(rng.uniform(0.2, 1.0)), int(rng.integers(0, 3)), bool(rng.integers(0, 2))))
    return pads


@pytest.mark.parametrize("ofmt", ["BGRA", "RGBA"])
@pytest.mark.parametrize("bg", ["checker", "transparent"])
def test_compositor_runs_of_pads(vfhip, metalref, ofmt, bg, monkeypatch):
    """RGBA / BGRA outputs are drawn in runs (comp_launch_runs): a launch per run of like pads, later runs in place over their
    bounding rectangle.  45 mixed pads on a frame wider than one wave: against the oracle, and bit for bit against the one-kernel
    path (VFHIP_COMP_ONE_PASS) and against the runs with the lean kernel switched off."""
    w, h = 600, 200
    pads = _mixed_pads(w, h, 45, 7)
    comp = vfhip.Compositor(0)
    comp.configure(ofmt, w, h)
    got = comp.composite(to_vf(vfhip, pads), background=bg)
    close(got, metalref.compositor(ofmt, w, h, pads, vfhip.BACKGROUNDS[bg]), f"runs {ofmt} {bg}", max_off_by_one=0.05)
    monkeypatch.setenv("VFHIP_COMP_ONE_PASS", "1")
    one = comp.composite(to_vf(vfhip, pads), background=bg)
    monkeypatch.delenv("VFHIP_COMP_ONE_PASS")
    assert np.array_equal(got, one), "runs differ from the single-kernel path"
    monkeypatch.setenv("VFHIP_COMP_NO_QUADS", "1")
    noq = comp.composite(to_vf(vfhip, pads), background=bg)
    monkeypatch.delenv("VFHIP_COMP_NO_QUADS")
    assert np.array_equal(got, noq), "lean kernel differs from the samplers' exact-texel path"
    monkeypatch.setenv("VFHIP_COMP_GENERAL", "1")
    gen = comp.composite(to_vf(vfhip, pads), background=bg)
    monkeypatch.delenv("VFHIP_COMP_GENERAL")
    assert np.array_equal(got, gen), "general kernel differs"
    monkeypatch.setenv("VFHIP_COMP_NO_420", "1")
    n420 = comp.composite(to_vf(vfhip, pads), background=bg)
    monkeypatch.delenv("VFHIP_COMP_NO_420")
    assert np.array_equal(got, n420), "the 4:2:0 row walker differs from the block kernel"
    monkeypatch.setenv("VFHIP_COMP_NO_COVER", "1")
    ncov = comp.composite(to_vf(vfhip, pads), background=bg)
    monkeypatch.delenv("VFHIP_COMP_NO_COVER")
    assert np.array_equal(got, ncov), "skipping what later opaque pads overwrite changed the picture"
    monkeypatch.setenv("VFHIP_COMP_NO_SCALED", "1")
    nsc = comp.composite(to_vf(vfhip, pads), background=bg)
    monkeypatch.delenv("VFHIP_COMP_NO_SCALED")
    assert np.array_equal(got, nsc), "the scaled-RGBA kernel differs from the general sampler"
    comp.close()


@pytest.mark.parametrize("fmt", ["NV12", "I420"])
@pytest.mark.parametrize("ofmt", ["BGRA", "RGBA"])
def test_compositor_420_mosaic(vfhip, metalref, fmt, ofmt, monkeypatch):
    """a mosaic of opaque 4:2:0 pads (k_compositor_420: first run over the frame with the later pads' rectangles skipped, the others in
    place without reading the target), a translucent and an additive one on top, odd sizes and pads hanging over every frame edge"""
    w, h = 520, 136
    rng = np.random.default_rng(3)
    geo = [(0, 0, 260, 68), (260, 0, 260, 68), (0, 68, 260, 68), (260, 68, 260, 68),          # four quadrants, opaque
           (100, 30, 161, 45), (-14, -6, 80, 50), (470, 100, 90, 60), (254, 60, 13, 17)]      # odd sizes, off-frame corners, a sliver across the seams
    pads = []
    for k, (x, y, pw, ph) in enumerate(geo):
        raw = smooth(fmt, pw, ph, 300 + k)
        alpha, blend = (1.0, 1) if k < 4 else [(0.6, 1), (1.0, 2), (1.0, 1), (0.8, 0)][k - 4]
        pads.append((fmt, pw, ph, raw, x, y, pw, ph, alpha, blend, bool(k & 1)))
    comp = vfhip.Compositor(0)
    comp.configure(ofmt, w, h)
    got = comp.composite(to_vf(vfhip, pads), background="checker")
    close(got, metalref.compositor(ofmt, w, h, pads, 0), f"4:2:0 mosaic {fmt} {ofmt}", max_off_by_one=0.05)
    for knob in ("VFHIP_COMP_NO_420", "VFHIP_COMP_NO_COVER", "VFHIP_COMP_ONE_PASS"):
        monkeypatch.setenv(knob, "1")
        other = comp.composite(to_vf(vfhip, pads), background="checker")
        monkeypatch.delenv(knob)
        assert np.array_equal(got, other), knob
    comp.close()


def test_compositor_runs_batch(vfhip, metalref):
    """a batch through the runs: lean run over the frame, NV12 run in place over its rectangle, lean run on top"""
    import torch
    ow, oh, n = 520, 96, 3
    a = [smooth("BGRA", 300, 96, 170 + k) for k in range(n)]
    b = [smooth("NV12", 64, 48, 180 + k) for k in range(n)]
    c = [smooth("RGBA", 37, 21, 190 + k) for k in range(n)]
    for f in a + c:
        f.reshape(-1, 4)[:, 3] = np.random.default_rng(5).integers(0, 256, f.size // 4)
    pa, pb, pc = ((x[0].size + 255) // 256 * 256 for x in (a, b, c))
    da, db, dc = _ring(a, pa), _ring(b, pb), _ring(c, pc)
    dout = torch.zeros((n, ow * oh * 4), dtype=torch.uint8, device="cuda")
    comp = vfhip.Compositor(0)
    comp.configure("RGBA", ow, oh)
    pads = [comp.pad("BGRA", 300, 96, da.data_ptr(), 0, 0, 300, 96, 0.9, "over"),
            comp.pad("BGRA", 300, 96, da.data_ptr(), 260, 0, 300, 96, 0.5, "add"),
            comp.pad("NV12", 64, 48, db.data_ptr(), 276, 20, 64, 48, 0.7, "over"),
            comp.pad("RGBA", 37, 21, dc.data_ptr(), 301, 33, 37, 21, 1.0, "over")]
    s = torch.cuda.Stream()
    torch.cuda.synchronize()
    comp.composite_device(pads, dout.data_ptr(), background="white", stream=s.cuda_stream, n_frames=n, pad_pitches=[pa, pa, pb, pc], out_pitch=ow * oh * 4)
    s.synchronize()
    out = dout.cpu().numpy()
    for k in range(n):
        want = metalref.compositor("RGBA", ow, oh, [("BGRA", 300, 96, a[k], 0, 0, 300, 96, 0.9, 1), ("BGRA", 300, 96, a[k], 260, 0, 300, 96, 0.5, 2),
                                                    ("NV12", 64, 48, b[k], 276, 20, 64, 48, 0.7, 1), ("RGBA", 37, 21, c[k], 301, 33, 37, 21, 1.0, 1)], 2)
        close(out[k], want, f"runs batch frame {k}")
    comp.close()


def test_compositor_many_opaque_tiles(vfhip, metalref, monkeypatch):
    """more opaque pads than a launch has cover rectangles (12 tiles of a 4 x 3 mosaic, mixed NV12 / I420 / BGRA-source, over a
    translucent full-frame pad that they hide): the skipping of overwritten areas must stay invisible"""
    w, h = 512, 192
    tw, th = 128, 64
    under = smooth("BGRA", w, h, 500)
    under.reshape(-1, 4)[:, 3] = np.random.default_rng(1).integers(0, 256, w * h)
    pads = [("BGRA", w, h, under, 0, 0, w, h, 0.8, 1, False)]
    for k in range(12):
        fmt = ["NV12", "I420", "BGRA"][k % 3]
        raw = smooth(fmt, tw, th, 510 + k)
        pads.append((fmt, tw, th, raw, (k % 4) * tw, (k // 4) * th, tw, th, 1.0, 0 if fmt == "BGRA" else 1, bool(k & 1)))
    comp = vfhip.Compositor(0)
    comp.configure("BGRA", w, h)
    got = comp.composite(to_vf(vfhip, pads), background="white")
    close(got, metalref.compositor("BGRA", w, h, pads, 2), "12 opaque tiles", max_off_by_one=0.05)
    for knob in ("VFHIP_COMP_NO_COVER", "VFHIP_COMP_NO_420", "VFHIP_COMP_NO_QUADS", "VFHIP_COMP_ONE_PASS"):
        monkeypatch.setenv(knob, "1")
        other = comp.composite(to_vf(vfhip, pads), background="white")
        monkeypatch.delenv(knob)
        assert np.array_equal(got, other), knob
    comp.close()


@pytest.mark.parametrize("ofmt", ["BGRA", "RGBA"])
def test_compositor_multiviewer_scaled_pads(vfhip, metalref, ofmt, monkeypatch):
    """feeds scaled into the tiles of a mosaic (k_compositor_scaled: RGBA / BGRA pads at any size), down- and up-scaled, odd
    positions, overhanging the frame, translucent with per-pixel alpha, all three operators; NV12 feeds scaled by the general kernel"""
    w, h = 640, 180
    rng = np.random.default_rng(11)
    pads = []
    geo = [(0, 0, 320, 90, 640, 360), (320, 0, 320, 90, 333, 201), (0, 90, 320, 90, 96, 54), (320, 90, 320, 90, 640, 360),
           (150, 40, 301, 77, 64, 48), (-33, -9, 120, 70, 50, 20), (600, 150, 90, 70, 200, 100), (317, 1, 7, 177, 3, 90)]
    for k, (x, y, dw, dh, pw, ph) in enumerate(geo):
        fmt = ["BGRA", "RGBA", "BGRA", "NV12", "RGBA", "BGRA", "I420", "RGBA"][k]
        raw = smooth(fmt, pw, ph, 700 + k)
        if fmt in ("BGRA", "RGBA") and k >= 4:
            raw.reshape(-1, 4)[:, 3] = rng.integers(0, 256, pw * ph)
        alpha, blend = [(1.0, 1), (1.0, 1), (0.9, 1), (1.0, 1), (0.7, 1), (1.0, 2), (0.5, 1), (1.0, 0)][k]
        pads.append((fmt, pw, ph, raw, x, y, dw, dh, alpha, blend, bool(k & 1)))
    comp = vfhip.Compositor(0)
    comp.configure(ofmt, w, h)
    got = comp.composite(to_vf(vfhip, pads), background="checker")
    close(got, metalref.compositor(ofmt, w, h, pads, 0), f"multiviewer {ofmt}", max_off_by_one=0.05)
    for knob in ("VFHIP_COMP_NO_SCALED", "VFHIP_COMP_GENERAL", "VFHIP_COMP_ONE_PASS", "VFHIP_COMP_NO_COVER"):
        monkeypatch.setenv(knob, "1")
        other = comp.composite(to_vf(vfhip, pads), background="checker")
        monkeypatch.delenv(knob)
        assert np.array_equal(got, other), knob
    comp.close()


def test_compositor_opaque_420_mosaic_batch(vfhip, metalref):
    """a batch of frames of opaque NV12 / I420 tiles (first run over the frame with the later tiles' rectangles skipped, the others in place
    without reading the target) plus a scaled BGRA inset: every frame of the batch against the oracle"""
    import torch
    ow, oh, n = 512, 128, 3
    tiles = [("NV12", 256, 64, 0, 0), ("I420", 256, 64, 256, 0), ("NV12", 256, 64, 0, 64), ("NV12", 256, 64, 256, 64)]
    frames = [[smooth(f, w, h, 900 + 10 * k + t) for k in range(n)] for t, (f, w, h, _, _) in enumerate(tiles)]
    inset = [smooth("BGRA", 96, 54, 950 + k) for k in range(n)]
    rings = [_ring(fr, (fr[0].size + 255) // 256 * 256) for fr in frames] + [_ring(inset, (inset[0].size + 255) // 256 * 256)]
    pitches = [r.shape[1] for r in rings]
    dout = torch.zeros((n, ow * oh * 4), dtype=torch.uint8, device="cuda")
    comp = vfhip.Compositor(0)
    comp.configure("BGRA", ow, oh)
    pads = [comp.pad(f, w, h, rings[t].data_ptr(), x, y, w, h, 1.0, "over", "bt709" if t & 1 else "bt601") for t, (f, w, h, x, y) in enumerate(tiles)]
    pads.append(comp.pad("BGRA", 96, 54, rings[4].data_ptr(), 200, 30, 144, 81, 0.8, "over"))
    s = torch.cuda.Stream()
    torch.cuda.synchronize()
    comp.composite_device(pads, dout.data_ptr(), background="checker", stream=s.cuda_stream, n_frames=n, pad_pitches=pitches, out_pitch=ow * oh * 4)
    s.synchronize()
    out = dout.cpu().numpy()
    for k in range(n):
        opads = [(f, w, h, frames[t][k], x, y, w, h, 1.0, 1, bool(t & 1)) for t, (f, w, h, x, y) in enumerate(tiles)]
        opads.append(("BGRA", 96, 54, inset[k], 200, 30, 144, 81, 0.8, 1, False))
        close(out[k], metalref.compositor("BGRA", ow, oh, opads, 0), f"opaque mosaic batch frame {k}", max_off_by_one=0.05)
    comp.close()


def test_overlay_jpeg_logo(vfhip, metalref, tmp_path):
    """a JPEG logo through the overlay loader (csrc/host_jpeg.hip): the frame equals the oracle's overlay of the pixels Pillow decodes"""
    Image = pytest.importorskip("PIL.Image")
    w, h = 96, 64
    rng = np.random.default_rng(4)
    pic = np.clip(np.stack(np.meshgrid(np.arange(40) * 6, np.arange(24) * 10), axis=-1).sum(-1)[..., None] / 2 + rng.normal(0, 20, (24, 40, 3)), 0, 255).astype(np.uint8)
    path = tmp_path / "logo.jpg"
    Image.fromarray(pic).save(path, quality=85, subsampling=2)
    with Image.open(path) as im:
        ref = np.asarray(im.convert("RGBA")).copy()            # opaque: premultiplication leaves it as it is
    ov = vfhip.Overlay(0)
    ov.configure("NV12", w, h)
    ov.load_image(str(path))
    assert ov.image_size == (40, 24)
    frame = smooth("NV12", w, h, 33)
    close(ov.process(frame, x=20, y=12, alpha=0.8), metalref.overlay("NV12", w, h, frame, "NV12", ref, x=20, y=12, alpha=0.8), "jpeg overlay")
    ov.close()


@pytest.mark.parametrize("ifmt", ["NV12", "I420"])
@pytest.mark.parametrize("ofmt", ["NV12", "I420", "BGRA", "RGBA"])
@pytest.mark.parametrize("w,h", [(96, 40), (8, 2), (200, 114), (12, 6), (1920, 1080)])
def test_videofilter_420_quad_kernel(vfhip, metalref, ifmt, ofmt, w, h, monkeypatch, vf_mode):
    """k_vf_point_420q (4 x 2 pixels per lane: the block's chroma neighbourhood as three window loads, dword / 16-byte stores) writes the
    bytes k_vf_point's 2 x 2 blocks write — EQUAL, not close: the same operations on the same inputs — and both sit on the oracle; edge
    lanes (clamped chroma columns and rows), the smallest frame and 1080p"""
    rng = np.random.default_rng(w + h)
    raw = rng.integers(0, 256, ol.raw_layout(ifmt, w, h)[1], dtype=np.uint8) if (w, h) != (1920, 1080) else smooth(ifmt, w, h, 3)
    vf = vfhip.VideoFilter(0)
    vf.configure(ifmt, w, h, ofmt, colorimetry="bt709")
    n = 5
    g = np.linspace(0, 1, n, dtype=np.float32)
    lut = np.ones((n, n, n, 4), np.float32)
    lut[..., 0] = g[None, None, :] * 0.8; lut[..., 1] = g[None, :, None] ** 1.3; lut[..., 2] = g[:, None, None]
    for name, kw, use_lut in (("identity", {}, False), ("colour", dict(brightness=0.1, contrast=1.2, saturation=1.3, hue=0.4, gamma=1.6, sepia=0.3, vignette=0.4), True),
                              ("sharpen", dict(sharpness=0.6, contrast=1.1), True), ("blur", dict(sharpness=-0.5), False)):   # k_vf_sharp: its region fill in quads
        prm = vfhip.filter_params(**kw)
        if use_lut:
            vf.set_lut(lut)
        else:
            vf.clear_lut()
        quad = vf.process(raw, prm)
        monkeypatch.setenv("VFHIP_VF_BLOCKS", "1")
        blocks = vf.process(raw, prm)
        monkeypatch.delenv("VFHIP_VF_BLOCKS")
        assert np.array_equal(quad, blocks), (name, int(np.abs(quad.astype(int) - blocks.astype(int)).max()))
        if (w, h) != (1920, 1080):
            vf_parity(vfhip, metalref, ifmt, ofmt, w, h, raw, kw, lut=lut if use_lut else None, m709=True, what=f"{vf_mode} {name} {ifmt}->{ofmt}", max_off_by_one=0.08, got=quad)
    vf.close()


@pytest.mark.parametrize("ifmt,ofmt", [("BGRA", "BGRA"), ("RGBA", "NV12"), ("NV12", "NV12"), ("I420", "BGRA"), ("NV12", "I420"), ("I420", "I420"), ("BGRA", "RGBA")])
@pytest.mark.parametrize("w,h", [(96, 54), (8, 2), (200, 114), (1920, 1080)])
def test_overlay_quad_kernel(vfhip, metalref, ifmt, ofmt, w, h, monkeypatch):
    """k_overlay_quad (4 x 2 pixels per lane) writes the bytes k_overlay's 2 x 2 blocks write — equal, not close — with and without an image,
    the image hanging over the frame's edges; and sits on the oracle"""
    rng = np.random.default_rng(w * 3 + h)
    raw = rng.integers(0, 256, ol.raw_layout(ifmt, w, h)[1], dtype=np.uint8) if (w, h) != (1920, 1080) else smooth(ifmt, w, h, 4)
    img = _logo(24, 16)
    ov = vfhip.Overlay(0)
    ov.configure(ifmt, w, h, ofmt, colorimetry="bt709")
    for kw in (None, dict(x=w / 3, y=h / 4, alpha=0.8), dict(x=-5, y=h - 9, width=50, height=30, alpha=1.0), dict(x=w - 10.5, y=-3.25, width=33.3, height=20.7, alpha=0.6)):
        if kw is None:
            ov.clear_image(); kw = {}
        else:
            ov.set_image(img)
        quad = ov.process(raw, **kw)
        monkeypatch.setenv("VFHIP_OV_BLOCKS", "1")
        blocks = ov.process(raw, **kw)
        monkeypatch.delenv("VFHIP_OV_BLOCKS")
        assert np.array_equal(quad, blocks), (kw, int(np.abs(quad.astype(int) - blocks.astype(int)).max()))
        if (w, h) != (1920, 1080):
            close(quad, metalref.overlay(ifmt, w, h, raw, ofmt, img if kw else None, m709=True, **kw), f"overlay {ifmt}->{ofmt} {kw}", max_off_by_one=0.03)
    ov.close()


@pytest.mark.parametrize("method", ["none", "clockwise", "rotate-180", "counterclockwise", "horizontal-flip", "vertical-flip", "upper-left-diagonal", "upper-right-diagonal"])
@pytest.mark.parametrize("ifmt,ofmt", [("BGRA", "BGRA"), ("RGBA", "BGRA")])
def test_transform_permutation_kernel(vfhip, metalref, method, ifmt, ofmt, monkeypatch):
    """RGB frames, no crop: k_transform_perm (the frame as a permutation of the input's pixels, chosen only when the host has proved that every tap of
    the sampler rounds back to one texel) writes the bytes the four-tap kernel writes — equal — for all eight methods, square and not, up to 2160p
    wide rows; with a crop, or a width that is not a multiple of 4, the four-tap kernel runs (same result either way by construction of the test)"""
    for (w, h, crop) in [(64, 48, (0, 0, 0, 0)), (48, 48, (0, 0, 0, 0)), (1920, 1080, (0, 0, 0, 0)), (3840, 16, (0, 0, 0, 0)), (64, 36, (3, 5, 7, 2)), (62, 36, (0, 0, 0, 0))]:
        raw = rnd(ifmt, w, h, 53)
        t = vfhip.Transform(0)
        t.configure(ifmt, w, h, ofmt)          # (the wrapper keeps the size: an axis-swapping method on a non-square frame scales — the four-tap kernel)
        perm = t.process(raw, method=method, crop=crop)
        monkeypatch.setenv("VFHIP_TR_GENERAL", "1")
        general = t.process(raw, method=method, crop=crop)
        monkeypatch.delenv("VFHIP_TR_GENERAL")
        assert np.array_equal(perm, general), (w, h, crop, int(np.abs(perm.astype(int) - general.astype(int)).max()))
        t.close()
