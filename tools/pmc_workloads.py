#!/usr/bin/env python3
"""tools/pmc_workloads.py — the element kernels of the BASELINE configs (and two neighbours) launched a few times each, as the command for the PMC passes
of tools/gpu_pmc_cmd.sh: C1 k_cs_taps_strip, C3 k_vf_sharp, the filter without sharpening (k_vf_point_rgba4), C4 k_compositor_quads + _420,
C5 k_deinterlace_420q (+ the headline kernel as its second leg), bicubic C2 k_cs_cubic_dot.  Rings as in bench_configs.py (beyond the Infinity Cache);
prints the algorithmic bytes per LAUNCH of every workload so that FETCH x 2 + WRITE per dispatch can be set against them."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "gstreamer-metal_amd"))
import torch  # noqa: E402
import vfhip  # noqa: E402
import bench_configs as bc  # noqa: E402

s = torch.cuda.Stream()
out = {}
for name, fn, kw in (("C1", bc.c1, {}), ("C3", bc.c3, {}), ("C4", bc.c4, {}), ("C5", bc.c5, dict(frames=64))):
    r = fn(torch, vfhip, s, 0, **kw)
    out[name] = {"kernel": r["kernel"], "frames_per_launch": r["frames_per_launch"], "algorithmic_bytes_per_launch": r["algorithmic_bytes_per_frame"] * r["frames_per_launch"],
                 "frames_per_s": r["frames_per_s"], "legs": r.get("legs")}
    torch.cuda.empty_cache()
# the filter without sharpening: colour stages + LUT
w, h, F = 1920, 1080, 64
fin, fout = bc._ring(torch, F, 4 * w * h, 1), torch.empty((F, 4 * w * h), dtype=torch.uint8, device="cuda")
vf = vfhip.VideoFilter(0)
vf.configure("BGRA", w, h)
prm, lut = bc.c3_params(vfhip)
prm.sharpness = 0.0
vf.set_lut(lut)
ms, n = bc._measure(torch, s, lambda: vf.process_device(fin.data_ptr(), fout.data_ptr(), prm, stream=s.cuda_stream, n_frames=F, in_pitch=fin.shape[1], out_pitch=fout.shape[1]))
out["C3 without sharpening"] = {"kernel": "k_vf_point_rgba4", "frames_per_launch": F, "algorithmic_bytes_per_launch": 2 * 4 * w * h * F, "frames_per_s": round(F / ms * 1e3, 1)}
vf.close()
del fin, fout
# bicubic C2
iw, ih, ow, oh, F = 3840, 2160, 1920, 1080, 32
size = vfhip.plane_layout("NV12", iw, ih)[1]
fin, fout = bc._ring(torch, F, size, 40), torch.empty((F, 4 * ow * oh), dtype=torch.uint8, device="cuda")
cs = vfhip.ConvertScale(0)
cs.configure("NV12", iw, ih, "BGRA", ow, oh, method="bicubic", colorimetry="bt2020", chroma_site="mpeg2")
ms, n = bc._measure(torch, s, lambda: cs.process_device(fin.data_ptr(), fout.data_ptr(), stream=s.cuda_stream, n_frames=F, in_pitch=fin.shape[1], out_pitch=fout.shape[1]))
out["C2 bicubic"] = {"kernel": cs.kernel_name, "frames_per_launch": F, "algorithmic_bytes_per_launch": (size + 4 * ow * oh) * F, "frames_per_s": round(F / ms * 1e3, 1)}
cs.close()
print(json.dumps(out), flush=True)
