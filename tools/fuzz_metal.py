#!/usr/bin/env python3
"""One-off fuzz of the `metal`-numerics kernels on the GPU against oracle/metalref.c (tolerance: 1 LSB): random formats, odd
sizes, parameters.  usage: fuzz_metal.py [cases] [seed] [element 0..5]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gstreamer-metal_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import oracle_lib as ol  # noqa: E402
import vfhip  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
mr = ol.load_metalref()
F6, F4 = ["BGRA", "RGBA", "NV12", "I420", "UYVY", "YUY2"], ["BGRA", "RGBA", "NV12", "I420"]
bad = 0


def frame(fmt, w, h):
    return rng.integers(0, 256, ol.raw_layout(fmt, w, h)[1], dtype=np.uint8)


def check(what, got, want, tol=1, rare=None, **record):
    """tol: the largest byte difference allowed; rare: the share of bytes that may exceed +-1 when tol > 1.  A mismatch is written out whole
    (inputs, both outputs) so that a one-off can be diagnosed from the record instead of being chased"""
    global bad
    d = np.abs(np.asarray(got).astype(int).reshape(-1) - np.asarray(want).astype(int).reshape(-1))
    if d.max() > tol or (rare is not None and (d > 1).mean() > rare):
        bad += 1
        dump_dir = os.path.join(ROOT, "gpurun_out", "fuzz_dumps")
        os.makedirs(dump_dir, exist_ok=True)
        dump = os.path.join(dump_dir, f"metal_seed{sys.argv[2] if len(sys.argv) > 2 else 1}_case{case}.npz")
        np.savez_compressed(dump, got=np.asarray(got).reshape(-1), want=np.asarray(want).reshape(-1), what=np.array([what]), **record)
        print("MISMATCH", what, "max", int(d.max()), "count", int((d > 1).sum()), "| record:", dump, flush=True)


ONLY = int(sys.argv[3]) if len(sys.argv) > 3 else -1                # restrict to one element (3 = compositor)
for case in range(N):
    kind = rng.integers(6) if ONLY < 0 else ONLY
    w, h = int(rng.integers(2, 150)), int(rng.integers(2, 110))
    if kind == 3 and rng.integers(2):
        w, h = 4 * int(rng.integers(1, 160)), int(rng.integers(2, 140))          # the compositor's run kernels need width % 4 == 0; wider than one wave
    m709 = bool(rng.integers(2))
    col = "bt709" if m709 else "bt601"
    if kind == 0:
        ifmt, ofmt = F6[rng.integers(6)], F6[rng.integers(6)]
        ow, oh = int(rng.integers(2, 150)), int(rng.integers(2, 110))
        linear, borders = bool(rng.integers(2)), bool(rng.integers(3) == 0)
        raw = frame(ifmt, w, h)
        cs = vfhip.ConvertScale(0)
        cs.configure(ifmt, w, h, ofmt, ow, oh, method="bilinear" if linear else "nearest", numerics="metal", colorimetry=col, add_borders=borders, border_color=0xFF204060)
        got = cs.process(raw)
        cs.close()
        check(f"convertscale {ifmt}{(w, h)}->{ofmt}{(ow, oh)} lin={linear} borders={borders}", got,
              mr.convertscale(ifmt, w, h, raw, ofmt, ow, oh, linear=linear, add_borders=borders, border=0xFF204060, m709_in=m709, m709_out=m709))
    elif kind == 1:
        fmt = F4[rng.integers(4)]
        method = ["bob", "weave", "linear", "greedyh"][rng.integers(4)]
        tff, thr = bool(rng.integers(2)), float(rng.random() * 0.3)
        d = vfhip.Deinterlace(0)
        d.configure(fmt, w, h, colorimetry=col)
        prev = None
        for _ in range(3):
            raw = frame(fmt, w, h)
            check(f"deinterlace {fmt}{(w, h)} {method} tff={tff}", d.process(raw, method=method, tff=tff, threshold=thr),
                  mr.deinterlace(fmt, w, h, raw, prev, vfhip.DEINTERLACE_METHODS[method], tff=tff, threshold=thr, m709=m709))
            prev = raw
        d.close()
    elif kind == 2:
        ifmt, ofmt = F4[rng.integers(4)], F4[rng.integers(4)]
        kw = dict(brightness=float(rng.uniform(-0.5, 0.5)), contrast=float(rng.uniform(0.3, 2)), saturation=float(rng.uniform(0, 2)), gamma=float(rng.uniform(0.3, 3)))
        if rng.integers(2): kw["hue"] = float(rng.uniform(-3, 3))
        if rng.integers(2): kw["sharpness"] = float(rng.uniform(-1, 1))
        if rng.integers(2): kw["sepia"] = float(rng.random())
        if rng.integers(2): kw["vignette"] = float(rng.random())
        if rng.integers(2): kw["invert"] = True
        if rng.integers(2): kw.update(chroma_key=(float(rng.random()), float(rng.random()), float(rng.random())), tolerance=float(rng.random() * 0.5), smoothness=float(0.05 + rng.random() * 0.3))
        raw = frame(ifmt, w, h)
        vf = vfhip.VideoFilter(0)
        vf.configure(ifmt, w, h, ofmt, colorimetry=col)
        prm = vfhip.filter_params(**kw)
        # sharpened frames: an off-by-one byte of pass 1 (legitimate) leaves the unsharp mask times 1 + 2 |amount| (tests/test_metal_elements_gpu.py, vf_parity)
        amount = abs(kw.get("sharpness", 0.0))
        check(f"videofilter {ifmt}->{ofmt}{(w, h)} {sorted(kw.items())}", vf.process(raw, prm), mr.videofilter(ifmt, w, h, raw, ofmt, ol.mr_filter_params(prm), m709=m709),
              tol=1 + int(np.ceil(2 * amount)) if amount > 0.001 else 1, rare=0.004 if amount > 0.001 else None, raw=raw)
        vf.close()
    elif kind == 3:
        ofmt = F4[rng.integers(4)]
        n = int(rng.integers(0, 7))
        pads, opads = [], []
        for _ in range(n):
            f = F4[rng.integers(4)]
            pw, ph = int(rng.integers(2, 90)), int(rng.integers(2, 70))
            r = frame(f, pw, ph)
            x, y = int(rng.integers(-40, w)), int(rng.integers(-40, h))
            if rng.integers(2):
                x, y = x & ~1, y & ~1                                # on the chroma grid (k_compositor_420) half of the time
            dw, dh = (pw, ph) if rng.integers(3) else (int(rng.integers(1, 120)), int(rng.integers(1, 90)))
            a, b = (1.0 if rng.integers(3) == 0 else float(rng.random())), ["source", "over", "add"][rng.integers(3)]     # alpha 1: opaque pads
            pads.append((f, pw, ph, r, x, y, dw, dh, a, b, col))
            opads.append((f, pw, ph, r, x, y, dw, dh, a, vfhip.BLEND_MODES[b], m709))
        bg = ["checker", "black", "white", "transparent"][rng.integers(4)]
        comp = vfhip.Compositor(0)
        comp.configure(ofmt, w, h, colorimetry=col)
        check(f"compositor ->{ofmt}{(w, h)} {n} pads bg={bg}", comp.composite(pads, background=bg), mr.compositor(ofmt, w, h, opads, vfhip.BACKGROUNDS[bg], m709_out=m709))
        comp.close()
    elif kind == 4:
        ifmt, ofmt = F4[rng.integers(4)], F4[rng.integers(4)]
        method = list(vfhip.TRANSFORM_METHODS)[rng.integers(8)]
        crop = tuple(int(rng.integers(0, min(12, (d - 1) // 2) + 1)) for d in (h, h, w, w))
        raw = frame(ifmt, w, h)
        t = vfhip.Transform(0)
        t.configure(ifmt, w, h, ofmt, colorimetry=col)
        check(f"transform {ifmt}->{ofmt}{(w, h)} {method} {crop}", t.process(raw, method=method, crop=crop),
              mr.transform(ifmt, w, h, raw, ofmt, vfhip.TRANSFORM_METHODS[method], crop=crop, m709=m709))
        t.close()
    else:
        ifmt, ofmt = F4[rng.integers(4)], F4[rng.integers(4)]
        iw, ih = int(rng.integers(1, 60)), int(rng.integers(1, 50))
        img = rng.integers(0, 256, (ih, iw, 4), dtype=np.uint8)
        kw = dict(x=float(rng.uniform(-30, w)), y=float(rng.uniform(-30, h)), width=float(rng.choice([0, rng.uniform(1, 100)])), height=float(rng.choice([0, rng.uniform(1, 80)])), alpha=float(rng.random()))
        raw = frame(ifmt, w, h)
        ov = vfhip.Overlay(0)
        ov.configure(ifmt, w, h, ofmt, colorimetry=col)
        ov.set_image(img)
        check(f"overlay {ifmt}->{ofmt}{(w, h)} img{(iw, ih)} {kw}", ov.process(raw, **kw), mr.overlay(ifmt, w, h, raw, ofmt, img, m709=m709, **kw))
        ov.close()
print("cases", N, "mismatches", bad)
sys.exit(1 if bad else 0)
