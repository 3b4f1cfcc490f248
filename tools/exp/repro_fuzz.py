#!/usr/bin/env python3
"""tools/exp/repro_fuzz.py — re-run single gst-exact cases the fuzzer flagged (random frames, several seeds): where do the bytes differ?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "gstreamer-metal_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib, vfhip
orc = oracle_lib.load()
CASES = [("NV12", 24, 76, "YUY2", 24, 76, "bicubic", "bt2020", "jpeg"), ("UYVY", 12, 19, "RGBA", 51, 65, "bilinear", "bt601", "jpeg")]
for (ifmt, w, h, ofmt, ow, oh, method, col, site) in CASES:
    for seed in range(6):
        rng = np.random.default_rng(seed)
        raw = rng.integers(0, 256, vfhip.plane_layout(ifmt, w, h)[1], dtype=np.uint8)
        want = np.asarray(orc.convertscale(ifmt, w, h, raw, col, site, method, ofmt, ow, oh)).reshape(-1)
        outs = []
        for rep in range(3):
            cs = vfhip.ConvertScale(0)
            cs.configure(ifmt, w, h, ofmt, ow, oh, method=method, colorimetry=col, chroma_site=site)
            got = np.asarray(cs.process(raw)).reshape(-1); k = cs.kernel_name; cs.close()
            outs.append(got)
        d = np.nonzero(outs[0] != want)[0]
        stable = all(np.array_equal(outs[0], o) for o in outs[1:])
        bpp = 4 if ofmt in ("RGBA", "BGRA") else 2
        print(ifmt, (w, h), ofmt, (ow, oh), method, k, "seed", seed, "diff bytes", len(d), "stable", stable,
              "first (row, col, byte)", [(int(i) // (ow * bpp), (int(i) % (ow * bpp)) // bpp, int(i) % bpp) for i in d[:6]],
              "got", outs[0][d[:6]].tolist(), "want", want[d[:6]].tolist(), flush=True)
