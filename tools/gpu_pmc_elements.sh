#!/bin/bash
# PMC passes (separate rocprofv3 runs per counter group) for the element kernels driven by tools/bench_elements.py main
set -o pipefail
TAG=${1:-pmce}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
B="python3 tools/bench_elements.py ${2:-main}"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pmc_stats -- $B > $OUT/stats.jsonl 2> $OUT/stats.err || { tail $OUT/stats.err; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $B > $OUT/pmc_fetch.jsonl 2> $OUT/pmc_fetch.err || { tail $OUT/pmc_fetch.err; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $B > $OUT/pmc_write.jsonl 2> $OUT/pmc_write.err || { tail $OUT/pmc_write.err; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS --output-format csv -d $OUT/pmc_sq -- $B > $OUT/pmc_sq.jsonl 2> $OUT/pmc_sq.err || { tail $OUT/pmc_sq.err; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_grbm -- $B > $OUT/pmc_grbm.jsonl 2> $OUT/pmc_grbm.err || { tail -5 $OUT/pmc_grbm.err; }
python tools/pmc_summary.py $OUT > /dev/null
find $OUT/pmc_stats -name '*kernel_stats.csv' | head -1 | xargs -r cut -c1-160 | grep vfhip
find $OUT -name '*kernel_trace.csv' -size +1M -delete; find $OUT -name '*counter_collection.csv' -size +4M -delete
python - <<PY
import json
d = json.load(open("$OUT/pmc_summary.json"))
for k, c in d.items():
    g = {n: v["avg"] for n, v in c.items()}
    line = [k[:60]]
    if "SQ_WAVES" in g and "SQ_INSTS_VALU" in g:
        line.append(f"valu/wave {g['SQ_INSTS_VALU'] / max(g['SQ_WAVES'], 1):.0f}")
    if "SQ_ACTIVE_INST_VALU" in g and "SQ_BUSY_CYCLES" in g:
        line.append(f"active_valu/busy {g['SQ_ACTIVE_INST_VALU'] / max(g['SQ_BUSY_CYCLES'], 1):.2f}")
    if "FETCH_SIZE" in g:
        line.append(f"fetch {g['FETCH_SIZE'] * 1024 * 2 / 1e6:.1f} MB(x2)")
    if "WRITE_SIZE" in g:
        line.append(f"write {g['WRITE_SIZE'] * 1024 / 1e6:.1f} MB")
    if "GRBM_GUI_ACTIVE" in g:
        line.append(f"gui_active {g['GRBM_GUI_ACTIVE']:.0f}")
    if "GRBM_GUI_ACTIVE" in g and "SQ_ACTIVE_INST_VALU" in g:
        # SQ_* count quad-cycles summed over the chip's 1024 SIMDs; GRBM_GUI_ACTIVE sums the 8 XCDs' busy cycles (MI355X_MICROARCH.md)
        line.append(f"valu_busy {g['SQ_ACTIVE_INST_VALU'] * 4 / (g['GRBM_GUI_ACTIVE'] / 8 * 1024):.2f}")
    print("  ".join(line))
PY
