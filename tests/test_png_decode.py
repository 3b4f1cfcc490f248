"""libvfhip's PNG decoder (host code; the overlay image and PNG LUT loaders sit on it): every colour type, 8 and 16 bit,
all five scan-line filters, split IDAT — against PNGs written by tests/png_util.py.  No GPU needed."""
import ctypes as C
import os

import numpy as np
import pytest

import png_util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "gstreamer-metal_amd", "libvfhip.so")


@pytest.fixture(scope="module")
def lib():
    l = C.CDLL(LIB)
    l.vfhip_image_decode_png.argtypes = [C.c_char_p, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    l.vfhip_image_free.argtypes = [C.POINTER(C.c_uint8)]
    l.vfhip_last_error_string.restype = C.c_char_p
    return l


def decode(lib, path):
    p, w, h = C.POINTER(C.c_uint8)(), C.c_int(), C.c_int()
    rc = lib.vfhip_image_decode_png(str(path).encode(), C.byref(p), C.byref(w), C.byref(h))
    if rc != 0:
        return rc, None
    out = np.ctypeslib.as_array(p, (h.value, w.value, 4)).copy()
    lib.vfhip_image_free(p)
    return 0, out


@pytest.mark.parametrize("ctype,ch", [(0, 1), (2, 3), (4, 2), (6, 4)])
@pytest.mark.parametrize("depth", [8, 16])
def test_colour_types_and_filters(lib, tmp_path, ctype, ch, depth):
    rng = np.random.default_rng(ctype * 10 + depth)
    w, h = 37, 23
    s = rng.integers(0, 1 << depth, (h, w, ch))
    path = tmp_path / "t.png"
    png_util.write_png(path, s, ctype, depth, filters=[0, 1, 2, 3, 4])
    rc, got = decode(lib, path)
    assert rc == 0, lib.vfhip_last_error_string()
    s8 = (s >> (depth - 8)).astype(np.uint8)
    want = np.full((h, w, 4), 255, np.uint8)
    if ctype == 0:
        want[..., :3] = s8[..., :1]
    elif ctype == 2:
        want[..., :3] = s8
    elif ctype == 4:
        want[..., :3] = s8[..., :1]
        want[..., 3] = s8[..., 1]
    else:
        want = s8
    assert np.array_equal(got, want)


def test_palette_with_transparency(lib, tmp_path):
    rng = np.random.default_rng(1)
    pal = rng.integers(0, 256, (16, 3))
    trns = rng.integers(0, 256, 7)                       # shorter than the palette: the rest is opaque
    idx = rng.integers(0, 16, (9, 31, 1))
    path = tmp_path / "p.png"
    png_util.write_png(path, idx, 3, 8, filters=[4, 0, 1], palette=pal, trns=trns)
    rc, got = decode(lib, path)
    assert rc == 0
    want = np.concatenate([pal[idx[..., 0]], np.where(idx < 7, np.concatenate([trns, np.full(9, 255)])[idx[..., 0]][..., None], 255)], axis=-1)
    assert np.array_equal(got, want.astype(np.uint8))


def test_refusals(lib, tmp_path):
    p = tmp_path / "i.png"
    png_util.write_png(p, np.zeros((4, 4, 3), np.uint8), 2, 8, interlace=1)
    assert decode(lib, p)[0] == -2 and b"interlaced" in lib.vfhip_last_error_string()
    q = tmp_path / "n.png"
    q.write_bytes(b"not a png at all, really" * 4)
    assert decode(lib, q)[0] == -2
    assert decode(lib, tmp_path / "missing.png")[0] == -1
    t = tmp_path / "t.png"
    png_util.write_png(t, np.zeros((8, 8, 4), np.uint8), 6)
    data = t.read_bytes()
    t.write_bytes(data[:len(data) - 40])                  # truncated stream
    assert decode(lib, t)[0] == -1
