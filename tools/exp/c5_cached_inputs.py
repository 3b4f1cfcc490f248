"""tools/exp/c5_cached_inputs.py — ablation: the greedy-H deinterlacer (NV12 2160p) with its inputs served from cache (batch input pitch 0:
every frame of the batch reads the same 12.4 MB), clocks sampled: how much of its time is memory?"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "gstreamer-metal_amd")); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, ROOT)
import torch
import vfhip
from bench_elements import ring, timed
from bench import ClockSampler
s = torch.cuda.Stream()
R, w, h = 24, 3840, 2160
size = vfhip.plane_layout("NV12", w, h)[1]
din, dout = ring(R, size, 3), ring(R, size, 4)
d = vfhip.Deinterlace(0)
d.configure("NV12", w, h)
dp = din.shape[1]
for name, ip, op in (("full: 24 distinct frames", dp, dp), ("inputs cached (in_pitch 0)", 0, dp), ("inputs and outputs cached", 0, 0)):
    def run():
        d.process_device(din.data_ptr(), dout.data_ptr(), method="greedyh", tff=True, threshold=0.1, stream=s.cuda_stream, n_frames=R, in_pitch=ip, out_pitch=op)
    t0 = time.time()
    while time.time() - t0 < 0.5:
        run()
    s.synchronize()
    clk = ClockSampler(0.02); clk.start()
    ms = timed(run, s, 200)
    c = clk.result() or {}
    print(json.dumps({"case": name, "us_per_frame": round(ms * 1e3 / R, 3), "frames_per_s": round(R / ms * 1e3, 1), "sclk_MHz": c.get("sclk_MHz"),
                      "socket_power_W": c.get("socket_power_W"), "kcycles_per_frame": round(ms * 1e-3 / R * (c.get("sclk_MHz") or 0) * 1e3, 2)}), flush=True)
