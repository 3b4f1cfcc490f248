import json, os, sys
sys.path.insert(0, "/root/repo/gstreamer-metal_amd"); sys.path.insert(0, "/root/repo/tools")
import torch, vfhip
from bench_elements import ring, timed
s = torch.cuda.Stream()
for (ifmt, w, h, ofmt, ow, oh) in [("NV12", 3840, 2160, "BGRA", 1920, 1080), ("NV12", 1920, 1080, "BGRA", 1920, 1080), ("NV12", 1920, 1080, "BGRA", 1280, 720), ("BGRA", 1920, 1080, "NV12", 1920, 1080), ("NV12", 1920, 1080, "NV12", 1280, 720)]:
    isz, osz = vfhip.plane_layout(ifmt, w, h)[1], vfhip.plane_layout(ofmt, ow, oh)[1]
    F = max(4, min(64, int(1.5e9 // (isz + osz))))
    din, dout = ring(F, isz, 1), ring(F, osz, 2)
    cs = vfhip.ConvertScale(0)
    cs.configure(ifmt, w, h, ofmt, ow, oh, method="bilinear", colorimetry="bt709", chroma_site="mpeg2", numerics="metal")
    def run():
        cs.process_device(din.data_ptr(), dout.data_ptr(), stream=s.cuda_stream, n_frames=F, in_pitch=din.shape[1], out_pitch=dout.shape[1])
    us = timed(run, s, 8) * 1e3 / F
    print(json.dumps({"case": f"metal {ifmt} {w}x{h} -> {ofmt} {ow}x{oh}", "kernel": cs.kernel_name, "us_per_frame": round(us, 2), "GBps": round((isz + osz) / us / 1e3, 1)}), flush=True)
    cs.close()
