#!/usr/bin/env python3
"""tools/exp/vf_fast_vs_oracle.py — the video filter's fast path (hardware transcendentals, folded uniforms, fp16 LUT cells) against oracle/metalref.c,
stage by stage: histogram of |difference| per byte.  Run once per mode: default (fast), VFHIP_VF_LUT32=1 (fast colour, fp32 table), VFHIP_VF_EXACT=1."""
import json
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "gstreamer-metal_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as ol  # noqa: E402
import vfhip  # noqa: E402


def lut_grade(n):
    g = np.linspace(0, 1, n, dtype=np.float32)
    lut = np.ones((n, n, n, 4), np.float32)
    lut[..., 0], lut[..., 1], lut[..., 2] = g[None, None, :] ** 1.05, g[None, :, None], g[:, None, None] ** 0.95
    return lut


def lut_wild(n, seed=3):
    rng = np.random.default_rng(seed)
    lut = np.ones((n, n, n, 4), np.float32)
    lut[..., :3] = rng.random((n, n, n, 3), dtype=np.float32)
    return lut


ALL = dict(brightness=0.1, contrast=1.2, saturation=0.8, hue=0.3 * math.pi, gamma=1.5, sepia=0.2, vignette=0.3, noise=0.0,
           invert=True, chroma_key=(0.0, 1.0, 0.0), tolerance=0.3, smoothness=0.1)
CASES = [("brightness+contrast+saturation", dict(brightness=0.1, contrast=1.2, saturation=0.8), None),
         ("hue", dict(hue=0.3 * math.pi), None), ("gamma 1.5", dict(gamma=1.5), None), ("gamma 0.45", dict(gamma=0.45), None),
         ("sepia+invert", dict(sepia=0.2, invert=True), None), ("chroma key", dict(chroma_key=(0.0, 1.0, 0.0), tolerance=0.3, smoothness=0.1), None),
         ("vignette", dict(vignette=0.3), None), ("noise", dict(noise=0.1), None),
         ("lut grade 33 only", dict(), lut_grade(33)), ("lut random 17 only", dict(), lut_wild(17)),
         ("all colour", ALL, None), ("all colour + lut grade 33", ALL, lut_grade(33)),
         ("sharpen .5 only", dict(sharpness=0.5), None), ("all + sharpen .5 + lut (C3)", dict(ALL, sharpness=0.5), lut_grade(33)),
         ("all + sharpen 1.0", dict(ALL, sharpness=1.0), None), ("all + blur -1.0", dict(ALL, sharpness=-1.0), None)]


def main():
    w, h = 640, 360
    rng = np.random.default_rng(11)
    raw = rng.integers(0, 256, 4 * w * h, dtype=np.uint8)
    metalref = ol.load_metalref()
    vf = vfhip.VideoFilter(0)
    vf.configure("BGRA", w, h)
    mode = "exact" if os.environ.get("VFHIP_VF_EXACT") else ("fast+lut32" if os.environ.get("VFHIP_VF_LUT32") else "fast")
    for name, kw, lut in CASES:
        prm = vfhip.filter_params(**kw)
        if lut is not None:
            vf.set_lut(lut)
        else:
            vf.clear_lut()
        got = vf.process(raw, prm).astype(int)
        want = metalref.videofilter("BGRA", w, h, raw, "BGRA", ol.mr_filter_params(prm), lut=lut).astype(int)
        d = np.abs(got - want)
        hist = np.bincount(d.ravel(), minlength=4)
        print(json.dumps({"mode": mode, "case": name, "max": int(d.max()), "off_by_1": round(float(hist[1]) / d.size, 6),
                          "off_by_2": round(float(hist[2]) / d.size, 7), "beyond_2": int(hist[3:].sum())}), flush=True)
    vf.close()


main()
