#!/usr/bin/env python3
"""Ablation of the C3 videofilter cost: which property groups cost what (kernel-only, BGRA 1080p, device-resident).
[case ...] on the command line restricts the run to those cases (PMC passes of one case: tools/gpu_pmc_cmd.sh)."""
import json
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gstreamer-metal_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import vfhip  # noqa: E402
from bench_elements import ring, timed  # noqa: E402

w, h, R = 1920, 1080, 24
s = torch.cuda.Stream()
fin, fout = ring(R, 4 * w * h, 1), ring(R, 4 * w * h, 2)
vf = vfhip.VideoFilter(0)
vf.configure("BGRA", w, h)
n = 33
g = np.linspace(0, 1, n, dtype=np.float32)
lut = np.ones((n, n, n, 4), np.float32)
lut[..., 0], lut[..., 1], lut[..., 2] = g[None, None, :] ** 1.05, g[None, :, None], g[:, None, None] ** 0.95
ALL = dict(brightness=0.1, contrast=1.2, saturation=0.8, hue=0.3 * math.pi, gamma=1.5, sepia=0.2, noise=0.1, vignette=0.3, invert=True,
           chroma_key=(0.0, 1.0, 0.0), tolerance=0.3, smoothness=0.1)
cases = [("identity", {}, False), ("gamma", dict(gamma=1.5), False), ("hue", dict(hue=0.3 * math.pi), False),
         ("noise+vignette+key", dict(noise=0.1, vignette=0.3, chroma_key=(0.0, 1.0, 0.0), tolerance=0.3, smoothness=0.1), False),
         ("colour-all", ALL, False), ("lut-only", {}, True), ("colour-all+lut", ALL, True),
         ("sharp-only", dict(sharpness=0.5), False), ("sharp+lut", dict(sharpness=0.5), True), ("all-15+lut", dict(ALL, sharpness=0.5), True)]
want = sys.argv[1:]
for name, kw, use_lut in cases:
    if want and name not in want:
        continue
    if use_lut:
        vf.set_lut(lut)
    else:
        vf.clear_lut()
    prm = vfhip.filter_params(**kw)

    def run():
        vf.process_device(fin.data_ptr(), fout.data_ptr(), prm, stream=s.cuda_stream, n_frames=R, in_pitch=fin.shape[1], out_pitch=fout.shape[1])
    ms = timed(run, s, 5) / R
    print(json.dumps({"case": name, "us_per_frame": round(ms * 1e3, 2), "frames_per_s": round(1e3 / ms, 1),
                      "frac_of_8TBps": round(2 * 4 * w * h / (ms * 1e-3) / 8e12, 4)}), flush=True)
vf.close()
