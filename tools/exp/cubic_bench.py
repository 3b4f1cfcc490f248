#!/usr/bin/env python3
"""tools/exp/cubic_bench.py — C2 with method=bicubic alone (kernel-only, 32-frame launches): frames/s and the kernel that ran; the command for PMC passes"""
import json
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "gstreamer-metal_amd"))
import torch  # noqa: E402
import vfhip  # noqa: E402
import bench_configs as bc  # noqa: E402
s = torch.cuda.Stream()
iw, ih, ow, oh, F = 3840, 2160, 1920, 1080, 32
size = vfhip.plane_layout("NV12", iw, ih)[1]
fin, fout = bc._ring(torch, F, size, 40), torch.empty((F, 4 * ow * oh), dtype=torch.uint8, device="cuda")
cs = vfhip.ConvertScale(0)
cs.configure("NV12", iw, ih, "BGRA", ow, oh, method="bicubic", colorimetry="bt2020", chroma_site="mpeg2")
ms, n = bc._measure(torch, s, lambda: cs.process_device(fin.data_ptr(), fout.data_ptr(), stream=s.cuda_stream, n_frames=F, in_pitch=fin.shape[1], out_pitch=fout.shape[1]))
print(json.dumps({"kernel": cs.kernel_name, "frames_per_s": round(F / ms * 1e3, 1), "us_per_frame": round(ms / F * 1e3, 2)}), flush=True)
