#!/usr/bin/env python3
"""Per-launch duration of the headline kernel over a few seconds of back-to-back launches (power / clock dynamics).
usage: clock_series.py [seconds] [frames] [const]   (const: a constant-byte input ring instead of random bytes — data toggling costs power)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gstreamer-metal_amd"))
import torch
import vfhip

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
F = int(sys.argv[2]) if len(sys.argv) > 2 else 128
W, H, OW, OH = 3840, 2160, 1920, 1080
_, in_size = vfhip.plane_layout("NV12", W, H)
in_pitch = (in_size + 255) // 256 * 256
out_pitch = OW * OH * 4
if len(sys.argv) > 3 and sys.argv[3] == "const":
    ring_in = torch.full((F, in_pitch), 0x5a, dtype=torch.uint8, device="cuda")
else:
    ring_in = torch.randint(0, 256, (F, in_pitch), dtype=torch.uint8, device="cuda")
ring_out = torch.empty((F, out_pitch), dtype=torch.uint8, device="cuda")
cs = vfhip.ConvertScale(0)
cs.configure("NV12", W, H, "BGRA", OW, OH, colorimetry="bt2020", chroma_site="mpeg2")
s = torch.cuda.Stream()
n = int(secs / (F * 4.6e-6))
ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
torch.cuda.synchronize()
ev[0].record(s)
for k in range(n):
    cs.process_device(ring_in.data_ptr(), ring_out.data_ptr(), stream=s.cuda_stream, n_frames=F, in_pitch=in_pitch, out_pitch=out_pitch)
    ev[k + 1].record(s)
torch.cuda.synchronize()
d = [ev[k].elapsed_time(ev[k + 1]) * 1e3 for k in range(n)]
t = 0.0
step = max(n // 60, 1)
for k in range(0, n, step):
    chunk = d[k:k + step]
    print(f"t={t / 1e6:7.3f}s  us/launch {sum(chunk) / len(chunk):8.1f}  fps {F * len(chunk) / sum(chunk) * 1e6:10.0f}")
    t += sum(chunk)
