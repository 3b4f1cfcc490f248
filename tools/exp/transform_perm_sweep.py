#!/usr/bin/env python3
"""tools/exp/transform_perm_sweep.py — k_transform_perm against the four-tap kernel (VFHIP_TR_GENERAL=1, read per launch) on random RGB frame sizes,
all eight methods: the two must write the same bytes wherever the host-side proof lets the permutation kernel run."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "gstreamer-metal_amd"))
import numpy as np
import vfhip
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
bad = 0
for case in range(N):
    w = int(rng.integers(1, 1025)) * 4 if rng.integers(4) else int(rng.integers(1, 300))
    h = int(rng.integers(1, 400)) if rng.integers(3) else w
    ifmt, ofmt = [("BGRA", "BGRA"), ("RGBA", "BGRA"), ("RGBA", "RGBA")][rng.integers(3)]
    raw = rng.integers(0, 256, w * h * 4, dtype=np.uint8)
    t = vfhip.Transform(0)
    t.configure(ifmt, w, h, ofmt)
    for m in vfhip.TRANSFORM_METHODS:
        os.environ.pop("VFHIP_TR_GENERAL", None)
        a = t.process(raw, method=m)
        os.environ["VFHIP_TR_GENERAL"] = "1"
        b = t.process(raw, method=m)
        if not np.array_equal(a, b):
            bad += 1
            print("MISMATCH", ifmt, ofmt, w, h, m, int((a != b).sum()), flush=True)
    t.close()
print("cases", N, "x 8 methods, mismatches", bad)
