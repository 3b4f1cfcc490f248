#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace csv: per-kernel durations and the gaps between consecutive launches.
usage: trace_gaps.py <dir with *_kernel_trace.csv> [name substring]"""
import csv
import glob
import statistics
import sys

d, sub = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "k_cs_nv12_half")
for path in sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True)):
    rows = [r for r in csv.DictReader(open(path)) if sub in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
    gap = [(int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3 for a, b in zip(rows, rows[1:])]
    gap = [g for g in gap if g < 1000]
    print(path)
    print(f"  launches {len(rows)}  duration us: median {statistics.median(dur):.1f} min {min(dur):.1f} max {max(dur):.1f}  first5 {[round(x, 1) for x in dur[:5]]}")
    if gap:
        print(f"  gaps us: median {statistics.median(gap):.1f} min {min(gap):.1f} max {max(gap):.1f}")
