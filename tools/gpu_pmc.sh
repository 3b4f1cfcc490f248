#!/bin/bash
# tools/gpu_pmc.sh — PMC passes for the bench kernel (separate rocprofv3 runs per counter group, as
# MI355X_MICROARCH.md §rocprofv3 PMC slots prescribes: FETCH_SIZE and WRITE_SIZE do not fit one pass).
set -o pipefail
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
B="python bench.py --steps 30 --warmup 10 --no-cpu-baseline"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $B > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.err || { tail $OUT/pmc_fetch.err; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $B > $OUT/pmc_write.json 2> $OUT/pmc_write.err || { tail $OUT/pmc_write.err; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INST_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq -- $B > $OUT/pmc_sq.json 2> $OUT/pmc_sq.err || { tail $OUT/pmc_sq.err; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INST_VMEM_RD SQ_INST_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU --output-format csv -d $OUT/pmc_grbm -- $B > $OUT/pmc_grbm.json 2> $OUT/pmc_grbm.err || { tail -5 $OUT/pmc_grbm.err; }
python tools/pmc_summary.py $OUT
