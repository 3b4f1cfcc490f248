#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc csv output for the vfhip kernels: per-dispatch averages per counter."""
import csv, glob, json, os, sys
out = sys.argv[1]
res = {}
for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc = {}
        for row in csv.DictReader(open(f)):
            k = row.get("Kernel_Name", "")
            if "vfhip" not in k:
                continue
            name = k.split("(")[0].replace("void ", "")
            key = (name, row["Counter_Name"])
            acc.setdefault(key, []).append(float(row["Counter_Value"]))
        for (name, c), v in acc.items():
            res.setdefault(name, {})[c] = {"avg": sum(v) / len(v), "n": len(v)}
print(json.dumps(res, indent=1))
json.dump(res, open(os.path.join(out, "pmc_summary.json"), "w"), indent=1)
