#!/usr/bin/env python3
"""tools/exp/deint_hash.py — sha256 over the deinterlacer's outputs for a fixed, seeded set of 4:2:0 inputs (formats x field order x methods x sizes,
random and smooth frames, three-frame sequences).  Run under two builds ($VFHIP_LIB, tools/exp/run_variants_cmd.sh) or two settings of
$VFHIP_DEINT_ROWS: equal digests = byte-identical kernels.  --quick: a subset of a few seconds (tests/test_c5_chain_gpu.py runs it per strip height)."""
import hashlib
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "gstreamer-metal_amd"))
import numpy as np  # noqa: E402
import vfhip  # noqa: E402

QUICK = "--quick" in sys.argv
rng = np.random.default_rng(20261005)
H = hashlib.sha256()
n = 0
for fmt in ("NV12", "I420"):
    for (w, h) in (((1920, 1080), (36, 10), (132, 50), (64, 70)) if QUICK else ((1920, 1080), (64, 16), (36, 10), (8, 2), (132, 50), (720, 576))):
        _, size = vfhip.plane_layout(fmt, w, h)
        smooth = (np.add.outer(np.arange(h + h // 2 + 8), np.arange(4 * w)) % 251).astype(np.uint8).ravel()[:size]
        seqs = [[rng.integers(0, 256, size, dtype=np.uint8) for _ in range(3)],
                [np.roll(smooth, 3 * k) for k in range(3)],
                [np.clip(smooth.astype(np.int16) + rng.integers(-6, 7, size), 0, 255).astype(np.uint8) for _ in range(3)]]
        for col in (("bt709",) if QUICK else ("bt601", "bt709")):
            for method in (("bob", "weave", "greedyh") if QUICK else ("bob", "weave", "linear", "greedyh")):
                for tff in (True, False):
                    for thr in ((0.1, 0.02, 0.5) if method == "greedyh" and not QUICK else (0.1,)):
                        for seq in (seqs[2:] if QUICK else seqs):
                            de = vfhip.Deinterlace(0).configure(fmt, w, h, col)
                            for f in seq:
                                H.update(de.process(f, method=method, tff=tff, threshold=thr).tobytes())
                                n += 1
                            de.close()
print("frames", n, "sha256", H.hexdigest())
