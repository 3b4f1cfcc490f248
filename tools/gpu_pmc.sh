#!/bin/bash
# tools/gpu_pmc.sh [tag] [frames] — PMC passes for the bench kernel (separate rocprofv3 runs per counter group, as
# MI355X_MICROARCH.md §rocprofv3 PMC slots prescribes: FETCH_SIZE and WRITE_SIZE do not fit one pass), then
# profiles/traffic_latest.json (keyed by the kernel source sha) from the summary.
set -o pipefail
TAG=${1:-r02}
FR=${2:-512}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
B="python3 bench.py --frames $FR --launches-per-step 1 --steps 12 --warmup 3 --precondition 0.1 --no-cpu-baseline --no-ceilings --no-others"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $B > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.err || { tail $OUT/pmc_fetch.err; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $B > $OUT/pmc_write.json 2> $OUT/pmc_write.err || { tail $OUT/pmc_write.err; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq -- $B > $OUT/pmc_sq.json 2> $OUT/pmc_sq.err || { tail $OUT/pmc_sq.err; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INST_VMEM_RD SQ_INST_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU --output-format csv -d $OUT/pmc_grbm -- $B > $OUT/pmc_grbm.json 2> $OUT/pmc_grbm.err || { tail -5 $OUT/pmc_grbm.err; }
python3 tools/pmc_summary.py $OUT > /dev/null
python3 tools/update_traffic.py $OUT/pmc_summary.json $FR $TAG | tee $OUT/traffic_latest.json
cp profiles/traffic_latest.json $OUT/traffic_latest.json
