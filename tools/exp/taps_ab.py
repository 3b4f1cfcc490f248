#!/usr/bin/env python3
"""tools/exp/taps_ab.py — the shapes k_cs_taps serves (bilinear down-scales outside the tile kernel's reach), batched and one frame per
launch; run once per VFHIP_TAPS_ROWS value (1 = one output row per lane, the old kernel)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "gstreamer-metal_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import vfhip
from bench_elements import ring, timed
s = torch.cuda.Stream()
CASES = [("NV12", 1920, 1080, 640, 480), ("NV12", 3840, 2160, 1280, 720), ("NV12", 1920, 1080, 640, 360), ("NV12", 3840, 2160, 1600, 900),
         ("I420", 1920, 1080, 1280, 720), ("I420", 1920, 1080, 640, 480), ("I420", 3840, 2160, 1280, 720), ("NV12", 1280, 720, 320, 180)]
for (ifmt, w, h, ow, oh) in CASES:
    isz, osz = vfhip.plane_layout(ifmt, w, h)[1], vfhip.plane_layout("BGRA", ow, oh)[1]
    for F in (64, 1):
        din, dout = ring(F, isz, 1), ring(F, osz, 2)
        cs = vfhip.ConvertScale(0)
        cs.configure(ifmt, w, h, "BGRA", ow, oh, method="bilinear", colorimetry="bt709", chroma_site="mpeg2")
        def run():
            cs.process_device(din.data_ptr(), dout.data_ptr(), stream=s.cuda_stream, n_frames=F, in_pitch=din.shape[1], out_pitch=dout.shape[1])
        ms = timed(run, s, 8 if F > 1 else 200)
        print(json.dumps({"rows": os.environ.get("VFHIP_TAPS_ROWS", "default"), "case": f"{ifmt} {w}x{h} -> BGRA {ow}x{oh}", "frames_per_launch": F, "kernel": cs.kernel_name,
                          "us_per_frame": round(ms * 1e3 / F, 2)}), flush=True)
        cs.close(); del din, dout
