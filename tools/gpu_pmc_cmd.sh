#!/bin/bash
# tools/gpu_pmc_cmd.sh <tag> <python script and arguments ...> — separate rocprofv3 passes (kernel stats, HBM bytes, SQ issue / wait counters, LDS
# counters, GRBM) of one command on the GPU box; per-kernel averages in gpurun_out/<tag>/pmc_summary.json and a one-line digest per kernel.
set -o pipefail
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
pass () { n=$1; shift; timeout -k 10 300 rocprofv3 --kernel-trace "$@" --output-format csv -d $OUT/pmc_$n -- python3 $CMD > $OUT/$n.out 2> $OUT/$n.err || { tail -5 $OUT/$n.err; return 1; }; }
CMD="$*"
pass stats --stats || exit 1
pass fetch --pmc FETCH_SIZE || exit 1
pass write --pmc WRITE_SIZE || exit 1
pass sq --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS || exit 1
pass wait --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM || exit 1
pass grbm --pmc GRBM_GUI_ACTIVE || true
python tools/pmc_summary.py $OUT > /dev/null
find $OUT/pmc_stats -name '*kernel_stats.csv' | head -1 | xargs -r cut -c1-200 | grep -i "vfhip\|Name" > $OUT/kernel_stats_digest.txt
find $OUT -name '*kernel_trace.csv' -size +1M -delete; find $OUT -name '*counter_collection.csv' -size +4M -delete
python3 - <<PY
import json
d = json.load(open("$OUT/pmc_summary.json"))
for k, c in d.items():
    g = {n: v["avg"] for n, v in c.items()}
    line = [k[:70]]
    w = max(g.get("SQ_WAVES", 1), 1)
    if "SQ_INSTS_VALU" in g: line.append(f"valu/wave {g['SQ_INSTS_VALU'] / w:.0f}")
    if "SQ_INSTS_LDS" in g: line.append(f"lds/wave {g['SQ_INSTS_LDS'] / w:.0f}")
    wc = g.get("SQ_WAVE_CYCLES")
    if wc:
        for n in ("SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM"):
            if n in g: line.append(f"{n[3:].lower()}/wave_cycles {g[n] / wc:.3f}")
    if "SQ_LDS_BANK_CONFLICT" in g and g.get("SQ_LDS_IDX_ACTIVE"): line.append(f"lds_conflict/idx_active {g['SQ_LDS_BANK_CONFLICT'] / g['SQ_LDS_IDX_ACTIVE']:.3f}")
    if "FETCH_SIZE" in g: line.append(f"fetch {g['FETCH_SIZE'] * 1024 * 2 / 1e6:.1f} MB(x2)")
    if "WRITE_SIZE" in g: line.append(f"write {g['WRITE_SIZE'] * 1024 / 1e6:.1f} MB")
    if "GRBM_GUI_ACTIVE" in g and "SQ_ACTIVE_INST_VALU" in g and "SQ_BUSY_CYCLES" in g:
        line.append(f"gui_active/8 {g['GRBM_GUI_ACTIVE'] / 8:.0f} cycles")
        # SQ_ACTIVE_INST_VALU: quad-cycles summed over the chip's 1024 SIMDs in which a VALU instruction was issuing
        line.append(f"valu_issue_share {g['SQ_ACTIVE_INST_VALU'] * 4 / (g['GRBM_GUI_ACTIVE'] / 8 * 1024):.2f}")
    print("  ".join(line))
PY
