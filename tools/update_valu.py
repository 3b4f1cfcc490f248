#!/usr/bin/env python3
"""tools/update_valu.py <valu_roofline.json> [more ...] — merge the per-kernel VALU figures of tools/valu_roofline.py (one or several PMC runs) into
profiles/valu_latest.json, keyed by the sha of the kernel sources they were measured on: bench.py attaches a kernel's figure to its JSON line only for
that exact source (like roofline.traffic).  valu_issue_share = SQ_INSTS_VALU x the average issue cycles of the kernel's opcode mix / all SIMD cycles of
the dispatch; it is a LOWER bound on sub-millisecond dispatches (the counter-derived clock reads high there, MI355X_MICROARCH.md)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

out = {"source_sha16": bench.csrc_sha16(), "source": [os.path.relpath(os.path.abspath(a), ROOT) for a in sys.argv[1:]], "kernels": {}}
for path in sys.argv[1:]:
    for k, v in json.load(open(path)).items():
        short = k.split("::")[-1].split("<")[0]
        out["kernels"][short] = {"pmc_name": k, "valu_wave_instructions_per_dispatch": v["valu_wave_instructions_per_dispatch"], "avg_issue_cycles": v["avg_issue_cycles"],
                                 "valu_issue_share": v["valu_issue_share"], "classes": v["classes"]}
json.dump(out, open(os.path.join(ROOT, "profiles", "valu_latest.json"), "w"), indent=1)
print(json.dumps({k: v["valu_issue_share"] for k, v in out["kernels"].items()}))
