#!/usr/bin/env python3
"""tools/bench_elem_sweep.py — deinterlace / videofilter / transform / overlay on common 1080p configurations, batched device paths:
microseconds per frame and algorithmic GB/s (in + out).  A survey to find cells far off the pace of their neighbours."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gstreamer-metal_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import torch
import vfhip
from bench_elements import ring, timed
s = torch.cuda.Stream()
w, h, F = 1920, 1080, 24


def rep(name, ms, nbytes):
    us = ms * 1e3 / F
    print(json.dumps({"case": name, "us_per_frame": round(us, 2), "frames_per_s": round(1e6 / us), "algorithmic_GBps": round(nbytes / us / 1e3, 1)}), flush=True)


for fmt in ("NV12", "I420", "BGRA"):
    size = vfhip.plane_layout(fmt, w, h)[1]
    din, dout = ring(F, size, 1), ring(F, size, 2)
    d = vfhip.Deinterlace(0); d.configure(fmt, w, h)
    for m in ("bob", "linear", "weave", "greedyh"):
        def run():
            d.process_device(din.data_ptr(), dout.data_ptr(), method=m, tff=True, threshold=0.1, stream=s.cuda_stream, n_frames=F, in_pitch=din.shape[1], out_pitch=dout.shape[1])
        rep(f"deinterlace {fmt} 1080p {m}", timed(run, s, 8), (3 if m in ("weave", "greedyh") else 2) * size)
    d.close()
for ifmt, ofmt in (("BGRA", "BGRA"), ("NV12", "NV12"), ("NV12", "BGRA"), ("I420", "I420")):
    isz, osz = vfhip.plane_layout(ifmt, w, h)[1], vfhip.plane_layout(ofmt, w, h)[1]
    din, dout = ring(F, isz, 1), ring(F, osz, 2)
    vf = vfhip.VideoFilter(0); vf.configure(ifmt, w, h, ofmt)
    for name, kw in (("identity", {}), ("brightness+contrast+saturation", dict(brightness=0.1, contrast=1.2, saturation=1.3)), ("sharpen", dict(sharpness=0.5))):
        prm = vfhip.filter_params(**kw)
        def run():
            vf.process_device(din.data_ptr(), dout.data_ptr(), prm, stream=s.cuda_stream, n_frames=F, in_pitch=din.shape[1], out_pitch=dout.shape[1])
        rep(f"videofilter {ifmt}->{ofmt} 1080p {name}", timed(run, s, 8), isz + osz)
    vf.close()
for fmt in ("NV12", "BGRA"):
    size = vfhip.plane_layout(fmt, w, h)[1]
    din, dout = ring(F, size, 1), ring(F, size, 2)
    t = vfhip.Transform(0); t.configure(fmt, w, h, fmt)
    for m in ("none", "horizontal-flip", "rotate-180", "vertical-flip"):
        def run():
            t.process_device(din.data_ptr(), dout.data_ptr(), method=m, stream=s.cuda_stream, n_frames=F, in_pitch=din.shape[1], out_pitch=dout.shape[1])
        rep(f"transform {fmt} 1080p {m}", timed(run, s, 8), 2 * size)
    t.close()
    ov = vfhip.Overlay(0); ov.configure(fmt, w, h, fmt)
    ov.set_image(np.random.default_rng(0).integers(0, 256, (256, 256, 4), dtype=np.uint8))
    def run():
        ov.process_device(din.data_ptr(), dout.data_ptr(), x=100.0, y=60.0, alpha=0.8, stream=s.cuda_stream, n_frames=F, in_pitch=din.shape[1], out_pitch=dout.shape[1])
    rep(f"overlay {fmt} 1080p 256x256 logo", timed(run, s, 8), 2 * size)
    ov.close()
