#!/usr/bin/env python3
"""tools/exp/others_bench.py [C1 C3 C4 C5 ...] — bench_configs' measurements on their own (A/B runs of one translation unit's build flags)"""
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
for name in (sys.argv[1:] or ["C1", "C3", "C4", "C5"]):
    r = getattr(bc, name.lower())(torch, vfhip, s, 0, **({"frames": 64} if name == "C5" else {}))
    print(name, r["frames_per_s"], r["frac"], {k: v["frac"] for k, v in (r.get("legs") or {}).items()}, flush=True)
    torch.cuda.empty_cache()
