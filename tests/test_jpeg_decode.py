"""libvfhip's JPEG decoder (host code; the overlay image loader sits on it) against Pillow / libjpeg-turbo: the decoder restates the IJG
algorithms (islow IDCT, fancy up-sampling, fixed-point colour conversion), so the pixels are expected to be EQUAL, not close.  No GPU needed."""
import ctypes as C
import io
import os

import numpy as np
import pytest

Image = pytest.importorskip("PIL.Image")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "gstreamer-metal_amd", "libvfhip.so")


@pytest.fixture(scope="module")
def lib():
    l = C.CDLL(LIB)
    l.vfhip_image_decode.argtypes = [C.c_char_p, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    l.vfhip_image_free.argtypes = [C.POINTER(C.c_uint8)]
    l.vfhip_last_error_string.restype = C.c_char_p
    return l


def decode(lib, path):
    p, w, h = C.POINTER(C.c_uint8)(), C.c_int(), C.c_int()
    rc = lib.vfhip_image_decode(str(path).encode(), C.byref(p), C.byref(w), C.byref(h))
    if rc != 0:
        return rc, None
    out = np.ctypeslib.as_array(p, (h.value, w.value, 4)).copy()
    lib.vfhip_image_free(p)
    return 0, out


def picture(w, h, seed, noise):
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w]
    img = np.stack([128 + 100 * np.sin(x / 7.0 + seed), 128 + 100 * np.cos(y / 5.0), (x * 3 + y * 5 + 40 * seed) % 256], axis=-1)
    img += rng.normal(0, noise, img.shape)
    img[h // 3:h // 2, w // 4:w // 2] = (250, 10, 30)                 # hard chroma edges: the up-sampling filters matter
    return np.clip(img, 0, 255).astype(np.uint8)


def reference(path):
    with Image.open(path) as im:
        return np.asarray(im.convert("RGBA"))


@pytest.mark.parametrize("subsampling", [0, 1, 2])                     # 4:4:4, 4:2:2, 4:2:0
@pytest.mark.parametrize("quality", [30, 75, 95, 100])
def test_ycbcr_equals_libjpeg(lib, tmp_path, subsampling, quality):
    for (w, h) in [(64, 48), (37, 23), (1, 1), (8, 8), (17, 9), (9, 17), (255, 3), (2, 130)]:
        p = tmp_path / f"s{subsampling}_q{quality}_{w}x{h}.jpg"
        Image.fromarray(picture(w, h, w + h, 12.0)).save(p, quality=quality, subsampling=subsampling)
        rc, got = decode(lib, p)
        assert rc == 0, lib.vfhip_last_error_string()
        ref = reference(p)
        assert got.shape == ref.shape
        assert np.array_equal(got, ref), (subsampling, quality, w, h, int(np.abs(got.astype(int) - ref.astype(int)).max()))


def test_greyscale_restart_intervals_optimised_tables(lib, tmp_path):
    g = picture(123, 77, 5, 20.0)[..., 0]
    p = tmp_path / "g.jpg"
    Image.fromarray(g, mode="L").save(p, quality=80)
    rc, got = decode(lib, p)
    assert rc == 0 and np.array_equal(got, reference(p))
    rgb = picture(200, 120, 7, 25.0)
    for k, kw in enumerate([dict(quality=85, optimize=True), dict(quality=60, subsampling=2, restart_marker_blocks=3),
                            dict(quality=90, subsampling=1, restart_marker_rows=1), dict(quality=50, subsampling=0, restart_marker_blocks=1, optimize=True)]):
        p = tmp_path / f"r{k}.jpg"
        try:
            Image.fromarray(rgb).save(p, **kw)
        except TypeError:
            pytest.skip("this Pillow cannot write restart markers")
        rc, got = decode(lib, p)
        assert rc == 0, (kw, lib.vfhip_last_error_string())
        assert np.array_equal(got, reference(p)), kw


@pytest.mark.parametrize("subsampling", [0, 1, 2])
@pytest.mark.parametrize("quality", [25, 75, 98])
def test_progressive_equals_libjpeg(lib, tmp_path, subsampling, quality):
    """SOF2: spectral selection and successive approximation (libjpeg's standard ten-scan script has DC and AC refinement scans and
    end-of-band runs across blocks), with and without restart markers and optimised tables"""
    for k, (w, h) in enumerate([(64, 48), (37, 23), (1, 1), (8, 8), (17, 9), (9, 17), (255, 3), (2, 130), (200, 120)]):
        p = tmp_path / f"p{subsampling}_q{quality}_{w}x{h}.jpg"
        kw = dict(quality=quality, subsampling=subsampling, progressive=True)
        if k % 3 == 1:
            kw["restart_marker_blocks"] = 2
        if k % 2:
            kw["optimize"] = True
        try:
            Image.fromarray(picture(w, h, w + h, 14.0)).save(p, **kw)
        except TypeError:
            kw.pop("restart_marker_blocks")
            Image.fromarray(picture(w, h, w + h, 14.0)).save(p, **kw)
        assert b"\xff\xc2" in p.read_bytes()
        rc, got = decode(lib, p)
        assert rc == 0, (kw, w, h, lib.vfhip_last_error_string())
        ref = reference(p)
        assert np.array_equal(got, ref), (kw, w, h, int(np.abs(got.astype(int) - ref.astype(int)).max()))
    g = tmp_path / f"g{quality}.jpg"
    Image.fromarray(picture(123, 77, 5, 20.0)[..., 0], mode="L").save(g, quality=quality, progressive=True)
    rc, got = decode(lib, g)
    assert rc == 0 and np.array_equal(got, reference(g))


def test_progressive_truncated_or_damaged(lib, tmp_path):
    p = tmp_path / "p.jpg"
    Image.fromarray(picture(96, 64, 2, 10.0)).save(p, quality=85, progressive=True)
    data = p.read_bytes()
    for n in (len(data) // 4, len(data) // 2, len(data) - 5):
        t = tmp_path / f"t{n}.jpg"
        t.write_bytes(data[:n])
        assert decode(lib, t)[0] < 0
    # a sequential frame header in front of progressive scans: the scan parameters are refused, nothing is read out of range
    b = bytearray(data)
    b[data.index(b"\xff\xc2") + 1] = 0xc0
    d = tmp_path / "d.jpg"
    d.write_bytes(bytes(b))
    assert decode(lib, d)[0] < 0


def test_refusals_and_corrupt_files(lib, tmp_path):
    rgb = picture(64, 64, 3, 10.0)
    c = tmp_path / "cmyk.jpg"
    Image.fromarray(np.dstack([rgb, rgb[..., :1]]), mode="CMYK").save(c, quality=80)
    assert decode(lib, c)[0] == -2 and b"CMYK" in lib.vfhip_last_error_string()
    ok = tmp_path / "ok.jpg"
    Image.fromarray(rgb).save(ok, quality=80)
    data = ok.read_bytes()
    for n in (2, 3, 10, 100, len(data) // 2):
        t = tmp_path / f"t{n}.jpg"
        t.write_bytes(data[:n])
        assert decode(lib, t)[0] < 0
    assert decode(lib, tmp_path / "missing.jpg")[0] == -1
    q = tmp_path / "n.jpg"
    q.write_bytes(b"\xff\xd8" + b"not really a jpeg" * 10)
    assert decode(lib, q)[0] < 0


def test_overlay_loader_takes_jpeg_and_png(lib, tmp_path):
    """vfhip_image_decode picks the decoder by the file's first bytes, whatever the extension"""
    rgb = picture(40, 30, 9, 5.0)
    a, b = tmp_path / "logo.dat", tmp_path / "logo2.dat"
    Image.fromarray(rgb).save(a, format="JPEG", quality=90)
    Image.fromarray(rgb).save(b, format="PNG")
    rc, j = decode(lib, a)
    assert rc == 0 and np.array_equal(j, reference(a))
    rc, pn = decode(lib, b)
    assert rc == 0 and np.array_equal(pn[..., :3], rgb) and (pn[..., 3] == 255).all()
