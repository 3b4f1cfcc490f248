#!/usr/bin/env python3
"""tools/exp/taps_vs_tile.py — NV12 down-scales the LDS tile kernel serves: run with VFHIP_NO_BILINEAR_TILE=1 to see k_cs_taps(_strip) on them."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "gstreamer-metal_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import vfhip
from bench_elements import ring, timed
s = torch.cuda.Stream()
CASES = [("NV12", 1920, 1080, 1280, 720), ("NV12", 3840, 2160, 2560, 1440), ("NV12", 1920, 1080, 1600, 900), ("NV12", 3840, 2160, 1920, 1200), ("NV12", 1920, 1080, 1024, 576),
         ("NV12", 1920, 1080, 1920, 720), ("NV12", 1920, 1080, 1280, 1080), ("UYVY", 1920, 1080, 1280, 720)]
for (ifmt, w, h, ow, oh) in CASES:
    isz, osz = vfhip.plane_layout(ifmt, w, h)[1], vfhip.plane_layout("BGRA", ow, oh)[1]
    F = 32
    din, dout = ring(F, isz, 1), ring(F, osz, 2)
    cs = vfhip.ConvertScale(0)
    cs.configure(ifmt, w, h, "BGRA", ow, oh, method="bilinear", colorimetry="bt709", chroma_site="mpeg2")
    def run():
        cs.process_device(din.data_ptr(), dout.data_ptr(), stream=s.cuda_stream, n_frames=F, in_pitch=din.shape[1], out_pitch=dout.shape[1])
    ms = timed(run, s, 8)
    print(json.dumps({"tile": "off" if os.environ.get("VFHIP_NO_BILINEAR_TILE") else "on", "case": f"{ifmt} {w}x{h} -> BGRA {ow}x{oh}", "kernel": cs.kernel_name, "us_per_frame": round(ms * 1e3 / F, 2)}), flush=True)
    cs.close(); del din, dout
