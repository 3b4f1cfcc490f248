#!/usr/bin/env python3
"""tools/bench_cs_sweep.py — vfhipconvertscale (gst-exact) on common conversions, batched device path: which kernel runs, microseconds per
frame, algorithmic GB/s.  A survey to find cells that are far off the pace of their neighbours."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gstreamer-metal_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import vfhip
from bench_elements import ring, timed
s = torch.cuda.Stream()
CASES = [("NV12", 3840, 2160, "BGRA", 1920, 1080, "bilinear"), ("NV12", 3840, 2160, "BGRA", 1280, 720, "bilinear"), ("NV12", 3840, 2160, "BGRA", 3840, 2160, "bilinear"),
         ("NV12", 1920, 1080, "BGRA", 1920, 1080, "bilinear"), ("NV12", 1920, 1080, "BGRA", 1280, 720, "bilinear"), ("NV12", 1920, 1080, "BGRA", 3840, 2160, "bilinear"),
         ("NV12", 1920, 1080, "BGRA", 640, 360, "bilinear"), ("I420", 1920, 1080, "BGRA", 1280, 720, "bilinear"), ("BGRA", 1920, 1080, "NV12", 1920, 1080, "bilinear"),
         ("BGRA", 1920, 1080, "NV12", 1280, 720, "bilinear"), ("BGRA", 3840, 2160, "NV12", 1920, 1080, "bilinear"), ("NV12", 1920, 1080, "NV12", 1280, 720, "bilinear"),
         ("NV12", 3840, 2160, "NV12", 1920, 1080, "bilinear"), ("NV12", 1920, 1080, "I420", 1920, 1080, "bilinear"), ("UYVY", 1920, 1080, "BGRA", 1920, 1080, "bilinear"),
         ("YUY2", 1920, 1080, "NV12", 1920, 1080, "bilinear"), ("BGRA", 1920, 1080, "BGRA", 1280, 720, "bilinear"), ("BGRA", 1920, 1080, "RGBA", 1920, 1080, "bilinear"),
         ("UYVY", 1920, 1080, "BGRA", 1280, 720, "bilinear"), ("I420", 1920, 1080, "BGRA", 3840, 2160, "bilinear"), ("NV12", 1280, 720, "BGRA", 1920, 1080, "bilinear"),
         ("NV12", 1920, 1080, "BGRA", 1280, 720, "nearest"), ("NV12", 1920, 1080, "BGRA", 1280, 720, "bicubic"), ("NV12", 1920, 1080, "NV12", 1280, 720, "bicubic")]
for (ifmt, w, h, ofmt, ow, oh, method) in CASES:
    isz, osz = vfhip.plane_layout(ifmt, w, h)[1], vfhip.plane_layout(ofmt, ow, oh)[1]
    F = max(4, min(64, int(1.5e9 // (isz + osz))))
    din, dout = ring(F, isz, 1), ring(F, osz, 2)
    cs = vfhip.ConvertScale(0)
    try:
        cs.configure(ifmt, w, h, ofmt, ow, oh, method=method, colorimetry="bt709", chroma_site="mpeg2")
    except vfhip.VfHipError as e:
        print(json.dumps({"case": f"{ifmt} {w}x{h} -> {ofmt} {ow}x{oh} {method}", "error": str(e)})); continue
    def run():
        cs.process_device(din.data_ptr(), dout.data_ptr(), stream=s.cuda_stream, n_frames=F, in_pitch=din.shape[1], out_pitch=dout.shape[1])
    ms = timed(run, s, 8)
    us = ms * 1e3 / F
    print(json.dumps({"case": f"{ifmt} {w}x{h} -> {ofmt} {ow}x{oh} {method}", "kernel": cs.kernel_name, "numerics": cs.numerics_in_effect if hasattr(cs, "numerics_in_effect") else None,
                      "us_per_frame": round(us, 2), "frames_per_s": round(1e6 / us), "algorithmic_GBps": round((isz + osz) / us / 1e3, 1)}), flush=True)
    cs.close(); del din, dout
