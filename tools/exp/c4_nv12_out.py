import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "gstreamer-metal_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch, vfhip
from bench_elements import ring, timed
s = torch.cuda.Stream()
ow, oh, NC = 3840, 2160, 16
quads = [ring(NC, 4 * 1920 * 1080, 10 + k) for k in range(4)]
nvq = [ring(NC, vfhip.plane_layout("NV12", 1920, 1080)[1], 30 + k) for k in range(4)]
nv = ring(NC, vfhip.plane_layout("NV12", 1280, 720)[1], 20)
for ofmt in ("NV12", "I420", "BGRA"):
    osz = vfhip.plane_layout(ofmt, ow, oh)[1]
    out = torch.empty((NC, osz), dtype=torch.uint8, device="cuda")
    comp = vfhip.Compositor(0); comp.configure(ofmt, ow, oh)
    qp = [comp.pad("BGRA", 1920, 1080, quads[k].data_ptr(), (k % 2) * 1920, (k // 2) * 1080, 1920, 1080, 0.9, "over") for k in range(4)]
    nvp = comp.pad("NV12", 1280, 720, nv.data_ptr(), (ow - 1280) // 2, (oh - 720) // 2, 1280, 720, 0.7, "over", colorimetry="bt709")
    nq = [comp.pad("NV12", 1920, 1080, nvq[k].data_ptr(), (k % 2) * 1920, (k // 2) * 1080, 1920, 1080, 1.0, "over", colorimetry="bt709") for k in range(4)]
    for name, pads, pitches in (("C4 pads", qp + [nvp], [quads[0].shape[1]] * 4 + [nv.shape[1]]), ("4 NV12 quadrants", nq, [nvq[0].shape[1]] * 4)):
        def run():
            comp.composite_device(pads, out.data_ptr(), background="black", stream=s.cuda_stream, n_frames=NC, pad_pitches=pitches, out_pitch=out.shape[1])
        ms = timed(run, s, 8) / NC
        print(json.dumps({"case": f"{name} -> {ofmt} 2160p", "us_per_frame": round(ms * 1e3, 2), "frames_per_s": round(1e3 / ms, 1)}), flush=True)
    comp.close()
