"""tools/exp/c4_trace.py <dir> — from a rocprofv3 --kernel-trace of tools/bench_c4_variants.py: the compositor launches in time order with
durations and the gaps between consecutive ones (first 40)."""
import csv, glob, sys
for path in sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)):
    rows = [r for r in csv.DictReader(open(path)) if "k_compositor" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    prev_end = None
    for r in rows[:40]:
        st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        name = r["Kernel_Name"].replace("vfhip::", "").replace("(vfhip::CompParams)", "").replace("void ", "")
        print(f"{name:32s} dur {(en - st) / 1e3:8.1f} us   gap before {(st - prev_end) / 1e3 if prev_end else 0:8.1f} us")
        prev_end = en
