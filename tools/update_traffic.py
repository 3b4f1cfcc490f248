#!/usr/bin/env python3
"""tools/update_traffic.py <pmc_summary.json> <frames_per_launch> [tag] — derive HBM bytes per frame of the headline
kernel from a rocprofv3 --pmc summary (FETCH_SIZE and WRITE_SIZE collected in SEPARATE passes, both in KiB per dispatch;
FETCH_SIZE doubled per MI355X_MICROARCH.md §HBM: gfx950 tallies a 128-byte request as 64 bytes) and write
profiles/traffic_latest.json keyed by the sha of the kernel source it was measured on (bench.py prints the figure only
for that exact source)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (kernel_source_sha16 only; no GPU touched)

summary, frames = json.load(open(sys.argv[1])), int(sys.argv[2])
tag = sys.argv[3] if len(sys.argv) > 3 else os.path.basename(os.path.dirname(os.path.abspath(sys.argv[1])))
name = next(k for k in summary if "k_cs_nv12_half" in k)
c = summary[name]
fetch = c["FETCH_SIZE"]["avg"] * 1024 * 2
write = c["WRITE_SIZE"]["avg"] * 1024
out = {
    "source": f"profiles/{tag}_pmc_summary.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, bench --frames {frames}; KiB per dispatch, FETCH x2 gfx950 correction)",
    "source_sha16": bench.kernel_source_sha16(),
    "kernel": name,
    "frames_per_launch": frames,
    "fetch_bytes_per_launch_corrected_x2": fetch,
    "write_bytes_per_launch": write,
    "hbm_bytes_per_frame": round((fetch + write) / frames, 1),
    "algorithmic_bytes_per_frame": bench.ALG_C2,
}
json.dump(out, open(os.path.join(ROOT, "profiles", "traffic_latest.json"), "w"), indent=1)
print(json.dumps(out))
