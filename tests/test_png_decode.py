"""libvfhip's PNG decoder (host code; the overlay image and PNG LUT loaders sit on it): every colour type and bit depth, Adam7,
all three tRNS forms, all five scan-line filters, split IDAT — against PNGs written by tests/png_util.py.  No GPU needed."""
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


def _expand(s, ctype, depth, palette=None, trns=None):
    """what the decoder must produce for samples `s` (h, w, ch)"""
    h, w, _ = s.shape
    want = np.full((h, w, 4), 255, np.uint8)
    if ctype == 3:
        pal = np.asarray(palette)
        want[..., :3] = pal[s[..., 0]]
        if trns is not None:
            t = np.concatenate([np.asarray(trns), np.full(256, 255)])
            want[..., 3] = t[s[..., 0]]
        return want
    s8 = (s >> (depth - 8)).astype(np.uint8) if depth >= 8 else (s * (255 // ((1 << depth) - 1))).astype(np.uint8)
    if ctype == 0:
        want[..., :3] = s8[..., :1]
        if trns is not None:
            want[..., 3] = np.where(s[..., 0] == trns, 0, 255)
    elif ctype == 2:
        want[..., :3] = s8
        if trns is not None:
            want[..., 3] = np.where((s == np.asarray(trns)).all(axis=-1), 0, 255)
    elif ctype == 4:
        want[..., :3] = s8[..., :1]
        want[..., 3] = s8[..., 1]
    else:
        want = s8
    return want


def _key_bytes(vals):
    return [b for v in np.atleast_1d(vals) for b in (int(v) >> 8, int(v) & 255)]


@pytest.mark.parametrize("interlace", [0, 1])
@pytest.mark.parametrize("ctype,ch,depth", [(0, 1, 1), (0, 1, 2), (0, 1, 4), (0, 1, 8), (0, 1, 16), (2, 3, 8), (2, 3, 16), (3, 1, 1), (3, 1, 2), (3, 1, 4), (3, 1, 8),
                                            (4, 2, 8), (4, 2, 16), (6, 4, 8), (6, 4, 16)])
def test_every_depth_and_adam7(lib, tmp_path, ctype, ch, depth, interlace):
    """every colour type x bit depth of the PNG specification, progressive and Adam7, with sizes that leave some of the
    seven passes empty or one pixel wide"""
    rng = np.random.default_rng(ctype * 100 + depth + interlace)
    for (w, h) in [(37, 23), (1, 1), (2, 3), (5, 1), (1, 9), (8, 8), (9, 4)]:
        n = min(1 << depth, 256) if ctype == 3 else 1 << depth
        s = rng.integers(0, n, (h, w, ch))
        pal = rng.integers(0, 256, (n, 3)) if ctype == 3 else None
        path = tmp_path / f"t{w}x{h}.png"
        png_util.write_png(path, s, ctype, depth, filters=[0, 1, 2, 3, 4], palette=pal, interlace=interlace)
        rc, got = decode(lib, path)
        assert rc == 0, lib.vfhip_last_error_string()
        assert np.array_equal(got, _expand(s, ctype, depth, pal)), (ctype, depth, interlace, w, h)


@pytest.mark.parametrize("interlace", [0, 1])
def test_colour_key_transparency(lib, tmp_path, interlace):
    """tRNS of the grey and RGB colour types: one sample value / one colour is fully transparent, compared at the file's depth
    (a 16-bit key must match both bytes)"""
    rng = np.random.default_rng(9)
    for depth in (1, 2, 4, 8, 16):
        s = rng.integers(0, min(1 << depth, 6), (11, 13, 1)) * (257 if depth == 16 else 1)
        key = int(s[3, 4, 0])
        p = tmp_path / f"g{depth}.png"
        png_util.write_png(p, s, 0, depth, filters=[1, 4], trns=_key_bytes(key), interlace=interlace)
        rc, got = decode(lib, p)
        assert rc == 0 and np.array_equal(got, _expand(s, 0, depth, trns=key))
        assert (got[..., 3] == 0).any() and (got[..., 3] == 255).any()
    for depth in (8, 16):
        s = rng.integers(0, 3, (11, 13, 3)) * (0x101 if depth == 16 else 1) + (0x1200 if depth == 16 else 0)
        key = s[5, 6]
        p = tmp_path / f"c{depth}.png"
        png_util.write_png(p, s, 2, depth, filters=[3, 2], trns=_key_bytes(key), interlace=interlace)
        rc, got = decode(lib, p)
        assert rc == 0 and np.array_equal(got, _expand(s, 2, depth, trns=key))
        assert (got[..., 3] == 0).any() and (got[..., 3] == 255).any()
    # 16-bit: a sample that differs from the key only in its LOW byte is opaque
    s = np.full((2, 2, 1), 0x1234)
    s[0, 0, 0] = 0x1235
    p = tmp_path / "low.png"
    png_util.write_png(p, s, 0, 16, trns=_key_bytes(0x1234))
    rc, got = decode(lib, p)
    assert rc == 0 and got[0, 0, 3] == 255 and got[1, 1, 3] == 0


def test_refusals(lib, tmp_path):
    p = tmp_path / "i.png"
    png_util.write_png(p, np.zeros((4, 4, 3), np.uint8), 2, 8, interlace=2)
    assert decode(lib, p)[0] == -2 and b"interlace" in lib.vfhip_last_error_string()
    for ctype, depth in [(2, 4), (3, 16), (4, 2), (6, 1), (0, 3)]:
        d = tmp_path / f"d{ctype}_{depth}.png"
        png_util.write_png(d, np.zeros((4, 4, {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]), np.uint8), ctype, 8)
        b = bytearray(d.read_bytes()); b[24] = depth; d.write_bytes(bytes(b))      # IHDR depth byte (the CRC is not checked)
        assert decode(lib, d)[0] == -2 and b"bit depth" in lib.vfhip_last_error_string()
    q = tmp_path / "n.png"
    q.write_bytes(b"not a png at all, really" * 4)
    assert decode(lib, q)[0] == -2
    assert decode(lib, tmp_path / "missing.png")[0] == -1
    t = tmp_path / "t.png"
    png_util.write_png(t, np.zeros((8, 8, 4), np.uint8), 6)
    data = t.read_bytes()
    t.write_bytes(data[:len(data) - 40])                  # truncated stream
    assert decode(lib, t)[0] == -1


def test_against_pillow(lib, tmp_path):
    """an independent implementation (Pillow / libpng's reader) reads the test writer's files — every <= 8-bit colour type, Adam7
    included, palette and colour-key transparency — to the same RGBA as libvfhip's decoder: the writer and the decoder do not merely
    agree with each other"""
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(21)
    n = 0
    for interlace in (0, 1):
        for ctype, ch, depth in [(0, 1, 1), (0, 1, 2), (0, 1, 4), (0, 1, 8), (2, 3, 8), (3, 1, 1), (3, 1, 2), (3, 1, 4), (3, 1, 8), (4, 2, 8), (6, 4, 8)]:
            for (w, h) in [(37, 23), (3, 2), (9, 9)]:
                lim = 1 << depth
                s = rng.integers(0, lim, (h, w, ch))
                pal = rng.integers(0, 256, (lim, 3)) if ctype == 3 else None
                trns = None
                if ctype == 3:
                    trns = rng.integers(0, 256, max(1, lim // 2))
                elif ctype == 0 and depth == 8:
                    trns = _key_bytes(int(s[0, 0, 0]))
                elif ctype == 2:
                    trns = _key_bytes(s[0, 0])
                path = tmp_path / f"pil_{interlace}_{ctype}_{depth}_{w}.png"
                png_util.write_png(path, s, ctype, depth, filters=[4, 3, 2, 1, 0], palette=pal, trns=trns, interlace=interlace)
                rc, got = decode(lib, path)
                assert rc == 0, lib.vfhip_last_error_string()
                with Image.open(path) as im:
                    ref = np.asarray(im.convert("RGBA"))
                assert np.array_equal(got, ref), (interlace, ctype, depth, w, h)
                n += 1
    assert n == 66
    # and the other direction: files written by Pillow's own encoder
    for mode, arr in [("L", rng.integers(0, 256, (19, 21), dtype=np.uint8)), ("RGB", rng.integers(0, 256, (19, 21, 3), dtype=np.uint8)),
                      ("RGBA", rng.integers(0, 256, (19, 21, 4), dtype=np.uint8)), ("LA", rng.integers(0, 256, (19, 21, 2), dtype=np.uint8)),
                      ("1", rng.integers(0, 2, (19, 21), dtype=np.uint8) * 255)]:
        path = tmp_path / f"from_pil_{mode}.png"
        im = Image.fromarray(arr, mode="L" if mode == "1" else mode)
        if mode == "1":
            im = im.convert("1")
        im.save(path, optimize=True)
        rc, got = decode(lib, path)
        assert rc == 0, lib.vfhip_last_error_string()
        with Image.open(path) as back:
            assert np.array_equal(got, np.asarray(back.convert("RGBA"))), mode
