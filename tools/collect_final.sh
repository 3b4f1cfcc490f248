#!/bin/bash
# tools/collect_final.sh <tag> — after `gpurun -- bash tools/gpu_final.sh <tag>`: copy the run's keyed figures and summaries from gpurun_out/ (scratch)
# into profiles/ (tracked): traffic_latest.json / valu_latest.json (sha-keyed, read by bench.py) and profiles/<tag>_*.
set -e
T=$1
cd "$(dirname "$0")/.."
cp gpurun_out/$T/valu_latest.json gpurun_out/$T/traffic_latest.json profiles/
for f in bench_driver.json bench.json; do cp gpurun_out/${T}_check/$f profiles/${T}_$f; done
cp gpurun_out/${T}_check/prof/runc/*kernel_stats.csv profiles/${T}_kernel_stats.csv
cp gpurun_out/${T}_pmc/pmc_summary.json profiles/${T}_pmc_summary.json
cp gpurun_out/${T}_pmce/pmc_summary.json profiles/${T}_pmc_elements_raw.json
cp gpurun_out/$T/pmc_elements_digest.txt profiles/${T}_pmc_elements_digest.txt
{ tail -2 gpurun_out/${T}_check/pytest_gpu.log; tail -2 gpurun_out/${T}_check/pytest_gpu_poison.log; } > profiles/${T}_pytest_gpu_tail.txt
python3 - <<PY
import json, sys
sys.path.insert(0, ".")
import bench
t = json.load(open("profiles/traffic_latest.json")); v = json.load(open("profiles/valu_latest.json"))
assert t["source_sha16"] == bench.kernel_source_sha16(), "traffic_latest.json is not keyed to these sources"
assert v["source_sha16"] == bench.csrc_sha16(), "valu_latest.json is not keyed to these sources"
print("keyed to this tree:", t["source_sha16"], v["source_sha16"])
PY
