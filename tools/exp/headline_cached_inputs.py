"""tools/exp/headline_cached_inputs.py — ablation: k_cs_nv12_half with its inputs served from cache.  The same launch (512 frames, NV12 2160p -> BGRA 1080p) with
in_pitch = 0 (every frame of the batch reads the SAME 12.4 MB input: L2 / MALL hits, no HBM reads) and with out_pitch = 0 as well;
if the kernel time does not move, the memory system is not what bounds it."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "gstreamer-metal_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import vfhip
from bench_elements import ring, timed
sys.path.insert(0, ROOT)
from bench import ClockSampler
import time
s = torch.cuda.Stream()
F = 512
isz, osz = 3840 * 2160 * 3 // 2, 1920 * 1080 * 4
din = ring(F, isz, 1); dout = torch.empty((F, osz), dtype=torch.uint8, device="cuda")
cs = vfhip.ConvertScale(0)
cs.configure("NV12", 3840, 2160, "BGRA", 1920, 1080, colorimetry="bt2020", chroma_site="mpeg2")
for name, ip, op in (("full: 512 distinct frames in, 512 out", din.shape[1], osz), ("inputs cached (in_pitch 0)", 0, osz),
                     ("inputs and outputs cached (both pitches 0)", 0, 0), ("outputs cached only (out_pitch 0)", din.shape[1], 0)):
    def run():
        cs.process_device(din.data_ptr(), dout.data_ptr(), stream=s.cuda_stream, n_frames=F, in_pitch=ip, out_pitch=op)
    t0 = time.time()
    while time.time() - t0 < 0.5:                              # pre-condition: clocks and power settle under THIS case's load
        run()
    s.synchronize()
    clk = ClockSampler(0.02); clk.start()
    ms = timed(run, s, 100)
    c = clk.result() or {}
    print(json.dumps({"case": name, "ms_per_launch": round(ms, 4), "frames_per_s": round(F / ms * 1e3, 1), "kernel": cs.kernel_name,
                      "sclk_MHz": c.get("sclk_MHz"), "socket_power_W": c.get("socket_power_W"),
                      "cycles_per_launch_M": round(ms * 1e-3 * (c.get("sclk_MHz") or 0), 2)}), flush=True)
