#!/usr/bin/env python3
"""tools/bench_c4_variants.py — where does BASELINE configs[3] (compositor 4 x BGRA 1080p + NV12 720p -> BGRA 2160p) spend its time?
The full configuration next to stripped ones (kernel-only, device-resident, batches of 8 output frames)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gstreamer-metal_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
import vfhip  # noqa: E402
from bench_elements import ring, timed  # noqa: E402

s = torch.cuda.Stream()
ow, oh, NC = 3840, 2160, 32
quads = [ring(NC, 4 * 1920 * 1080, 10 + k) for k in range(4)]
nvs = vfhip.plane_layout("NV12", 1280, 720)[1]
nv = ring(NC, nvs, 20)
out = torch.empty((NC, 4 * ow * oh), dtype=torch.uint8, device="cuda")
comp = vfhip.Compositor(0)
comp.configure("BGRA", ow, oh)
qp = [comp.pad("BGRA", 1920, 1080, quads[k].data_ptr(), (k % 2) * 1920, (k // 2) * 1080, 1920, 1080, 0.9, "over") for k in range(4)]
qsrc = [comp.pad("BGRA", 1920, 1080, quads[k].data_ptr(), (k % 2) * 1920, (k // 2) * 1080, 1920, 1080, 1.0, "source") for k in range(4)]
nvp = comp.pad("NV12", 1280, 720, nv.data_ptr(), (ow - 1280) // 2, (oh - 720) // 2, 1280, 720, 0.7, "over", colorimetry="bt709")
nvodd = comp.pad("NV12", 1280, 720, nv.data_ptr(), (ow - 1280) // 2 + 1, (oh - 720) // 2 + 1, 1280, 720, 0.7, "over", colorimetry="bt709")
cases = [("C4 full: 4 quadrants (over, .9) + NV12 720p centred", qp + [nvp], [quads[0].shape[1]] * 4 + [nv.shape[1]], "black"),
         ("4 quadrants only (over, .9)", qp, [quads[0].shape[1]] * 4, "black"),
         ("4 quadrants, operator source, alpha 1", qsrc, [quads[0].shape[1]] * 4, "black"),
         ("4 quadrants over a checker background", qp, [quads[0].shape[1]] * 4, "checker"),
         ("no pads: background only", [], [], "black"),
         ("C4 with the NV12 pad one pixel off the chroma grid (general sampler path)", qp + [nvodd], [quads[0].shape[1]] * 4 + [nv.shape[1]], "black")]
# mosaics of camera-style pads: four NV12 / I420 1080p quadrants (every pad goes through the 4:2:0 sampler)
nvq = [ring(NC, vfhip.plane_layout("NV12", 1920, 1080)[1], 30 + k) for k in range(4)]
for fmt in ("NV12", "I420"):
    qq = [comp.pad(fmt, 1920, 1080, nvq[k].data_ptr(), (k % 2) * 1920, (k // 2) * 1080, 1920, 1080, 1.0, "over", colorimetry="bt709") for k in range(4)]
    cases.append((f"4 {fmt} 1080p quadrants (over, 1.0)", qq, [nvq[0].shape[1]] * 4, "black"))
# a multiviewer: four 1080p feeds scaled down to the quadrants of a 1080p output (every pad through the scaling sampler)
mv_out = torch.empty((NC, 4 * 1920 * 1080), dtype=torch.uint8, device="cuda")
MV = {}
for fmt, src in (("BGRA", quads), ("NV12", nvq)):
    qq = [comp.pad(fmt, 1920, 1080, src[k].data_ptr(), (k % 2) * 960, (k // 2) * 540, 960, 540, 1.0, "over", colorimetry="bt709") for k in range(4)]
    MV[f"multiviewer: 4 {fmt} 1080p feeds scaled to 960x540 quadrants of a 1080p output"] = (qq, [src[0].shape[1]] * 4)
for name, pads, pitches, bg in cases:
    def run():
        comp.composite_device(pads, out.data_ptr(), background=bg, stream=s.cuda_stream, n_frames=NC, pad_pitches=pitches, out_pitch=out.shape[1])
    ms = timed(run, s, 10) / NC
    print(json.dumps({"case": name, "us_per_frame": round(ms * 1e3, 2), "frames_per_s": round(1e3 / ms, 1)}), flush=True)
comp.close()
comp = vfhip.Compositor(0)
comp.configure("BGRA", 1920, 1080)
for name, (pads, pitches) in MV.items():
    def run():
        comp.composite_device(pads, mv_out.data_ptr(), background="black", stream=s.cuda_stream, n_frames=NC, pad_pitches=pitches, out_pitch=mv_out.shape[1])
    ms = timed(run, s, 10) / NC
    print(json.dumps({"case": name, "us_per_frame": round(ms * 1e3, 2), "frames_per_s": round(1e3 / ms, 1)}), flush=True)
comp.close()
