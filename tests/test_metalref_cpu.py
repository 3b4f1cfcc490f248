"""Self-consistency of the float oracle (oracle/metalref.c).  PARITY UNPINNED: nothing here (or anywhere) compares with
real Metal output — the reference cannot run on Linux and its tests hold no pixels.  These checks pin
the properties Appendix B of SURVEY.md states, so that the GPU tests compare against a sane restatement."""
import ctypes as C

import numpy as np

import oracle_lib as ol


def rnd(fmt, w, h, seed=0):
    return np.random.default_rng(seed).integers(0, 256, ol.raw_layout(fmt, w, h)[1], dtype=np.uint8)


def fparams(**kw):
    d = dict(brightness=0.0, contrast=1.0, saturation=1.0, hue=0.0, gamma=1.0, sharpness=0.0, sepia=0.0, noise=0.0,
             vignette=0.0, invert=0, chroma_key_enabled=0, key_r=0.0, key_g=0.0, key_b=0.0, key_tolerance=0.2,
             key_smoothness=0.1, frame_index=0)
    d.update(kw)
    return ol.MrFilterParams(**d)


def test_convertscale_identity_and_swizzle(metalref):
    raw = rnd("RGBA", 33, 17)
    assert np.array_equal(metalref.convertscale("RGBA", 33, 17, raw, "RGBA", 33, 17), raw)
    bgra = metalref.convertscale("RGBA", 33, 17, raw, "BGRA", 33, 17)
    assert np.array_equal(bgra.reshape(-1, 4)[:, [2, 1, 0, 3]], raw.reshape(-1, 4))
    near = metalref.convertscale("RGBA", 33, 17, raw, "RGBA", 66, 34, linear=False)
    assert np.array_equal(near.reshape(34, 66, 4)[::2, ::2], raw.reshape(17, 33, 4))


def test_convertscale_letterbox_border(metalref):
    raw = np.full(ol.raw_layout("BGRA", 64, 16)[1], 200, np.uint8)
    out = metalref.convertscale("BGRA", 64, 16, raw, "BGRA", 32, 32, add_borders=True, border=0xFF102030).reshape(32, 32, 4)
    assert (out[0, 0] == [0x30, 0x20, 0x10, 0xFF]).all() and (out[16, 16] == 200).all()
    assert (out[11, 5] == [0x30, 0x20, 0x10, 0xFF]).all() and (out[12, 5] == 200).all() and (out[19, 5] == 200).all() and (out[20, 5][3] == 0xFF)


def test_yuv_known_answers(metalref):
    for m709 in (False, True):
        raw = np.zeros(ol.raw_layout("NV12", 4, 4)[1], np.uint8)
        raw[:16] = 235
        raw[16:] = 128
        out = metalref.convertscale("NV12", 4, 4, raw, "BGRA", 4, 4, m709_in=m709)
        assert (out == 255).all()            # float matrix: white maps to 255 (GStreamer's integer path gives 253)
        raw[:16] = 16
        assert (metalref.convertscale("NV12", 4, 4, raw, "BGRA", 4, 4, m709_in=m709).reshape(-1, 4)[:, :3] == 0).all()


def test_deinterlace_properties(metalref):
    w, h = 16, 12
    cur, prev = rnd("RGBA", w, h, 1), rnd("RGBA", w, h, 2)
    c, p = cur.reshape(h, w, 4), prev.reshape(h, w, 4)
    bob = metalref.deinterlace("RGBA", w, h, cur, None, ol.C.c_int(0).value).reshape(h, w, 4)
    assert np.array_equal(bob[0::2], c[0::2])                                   # kept field copied
    exp = np.rint((c[0:-2:2].astype(np.float64) + c[2::2]) / 2)                 # ties to even
    assert np.abs(bob[1:-1:2].astype(int) - exp).max() <= 1
    last = (c[h - 2].astype(np.float64) + c[h - 1]) / 2                          # last line: below clamps to the line itself
    assert np.abs(bob[h - 1].astype(int) - np.rint(last)).max() <= 1
    assert np.array_equal(metalref.deinterlace("RGBA", w, h, cur, None, 2), bob.reshape(-1))   # linear == bob
    assert np.array_equal(metalref.deinterlace("RGBA", w, h, cur, None, 1), bob.reshape(-1))   # weave w/o history == bob
    weave = metalref.deinterlace("RGBA", w, h, cur, prev, 1, tff=False).reshape(h, w, 4)
    assert np.array_equal(weave[1::2], c[1::2]) and np.array_equal(weave[0::2], p[0::2])
    g0 = metalref.deinterlace("RGBA", w, h, cur, prev, 3, threshold=10.0).reshape(h, w, 4)   # everything "static" -> weave
    assert np.array_equal(g0[1::2], p[1::2])
    g1 = metalref.deinterlace("RGBA", w, h, cur, prev, 3, threshold=0.0)                      # everything "moving" -> bob
    assert np.array_equal(g1, bob.reshape(-1))


def test_videofilter_defaults_are_identity(metalref):
    raw = rnd("BGRA", 40, 24)
    out = metalref.videofilter("BGRA", 40, 24, raw, "BGRA", fparams())
    assert np.array_equal(out, raw)
    inv = metalref.videofilter("BGRA", 40, 24, raw, "BGRA", fparams(invert=1)).reshape(-1, 4)
    assert np.array_equal(inv[:, :3], 255 - raw.reshape(-1, 4)[:, :3]) and np.array_equal(inv[:, 3], raw.reshape(-1, 4)[:, 3])


def test_videofilter_identity_lut_and_blur(metalref):
    raw = rnd("RGBA", 40, 24)
    n = 17
    g = np.linspace(0, 1, n, dtype=np.float32)
    lut = np.ones((n, n, n, 4), np.float32)
    lut[..., 0] = g[None, None, :]
    lut[..., 1] = g[None, :, None]
    lut[..., 2] = g[:, None, None]
    out = metalref.videofilter("RGBA", 40, 24, raw, "RGBA", fparams(), lut=lut)
    assert np.abs(out.astype(int) - raw.astype(int)).max() <= 1
    flat = np.full_like(raw, 90)
    assert np.array_equal(metalref.videofilter("RGBA", 40, 24, flat, "RGBA", fparams(sharpness=0.7)), flat)      # blur of a constant
    assert np.abs(metalref.videofilter("RGBA", 40, 24, flat, "RGBA", fparams(sharpness=-0.7)).astype(int) - 90).max() <= 1


def test_compositor_properties(metalref):
    w, h = 32, 24
    chk = metalref.compositor("RGBA", w, h, [], 0).reshape(h, w, 4)
    assert set(np.unique(chk[..., 0])) == {128, 191} and chk[0, 0, 0] == 128 and chk[0, 8, 0] == 191 and chk[8, 8, 0] == 128
    assert (metalref.compositor("RGBA", w, h, [], 1).reshape(-1, 4) == [0, 0, 0, 255]).all()
    assert (metalref.compositor("RGBA", w, h, [], 3) == 0).all()
    src = rnd("RGBA", w, h, 3)
    src.reshape(-1, 4)[:, 3] = 255
    out = metalref.compositor("RGBA", w, h, [("RGBA", w, h, src, 0, 0, w, h, 1.0, 1)], 1)
    assert np.array_equal(out, src)                                                # opaque OVER at 1:1 == copy
    part = metalref.compositor("RGBA", w, h, [("RGBA", w, h, src, 8, 4, 16, 8, 1.0, 0)], 2).reshape(h, w, 4)
    assert (part[0, 0] == 255).all() and (part[3, 8] == 255).all() and not (part[4:12, 8:24] == 255).all()
    half = metalref.compositor("RGBA", w, h, [("RGBA", w, h, np.full_like(src, 255), 0, 0, w, h, 0.5, 1)], 1).reshape(-1, 4)
    assert (np.abs(half[:, :3].astype(int) - 128) <= 1).all() and (half[:, 3] == 255).all()     # 0.5*1 + 0*(1-.5); alpha .5 + 1*.5


def test_transform_properties(metalref):
    w = h = 16
    raw = rnd("RGBA", w, h, 9)
    img = raw.reshape(h, w, 4).astype(int)
    ident = metalref.transform("RGBA", w, h, raw, "RGBA", 0).reshape(h, w, 4).astype(int)
    assert np.abs(ident - img).max() <= 1                                   # texel centres: linear sampler ~ exact texel
    hf = metalref.transform("RGBA", w, h, raw, "RGBA", 4).reshape(h, w, 4).astype(int)
    assert np.abs(hf - img[:, ::-1]).max() <= 1
    r180 = metalref.transform("RGBA", w, h, raw, "RGBA", 2).reshape(h, w, 4).astype(int)
    assert np.abs(r180 - img[::-1, ::-1]).max() <= 1
    crop = metalref.transform("RGBA", w, h, np.full_like(raw, 200), "RGBA", 0, (4, 4, 4, 4))
    assert (crop == 200).all()                                               # cropping a constant frame zooms, never leaves the image
