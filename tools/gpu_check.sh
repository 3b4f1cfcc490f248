#!/bin/bash
# tools/gpu_check.sh — one gpurun call: GPU parity tests, smoke, bench line, rocprofv3 kernel stats.
# usage: gpurun --timeout 900 -- 'bash tools/gpu_check.sh [tag]'
set -o pipefail
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1 || { tail -20 $OUT/smoke.log; exit 1; }
tail -2 $OUT/smoke.log
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; rc=$?
tail -15 $OUT/pytest_gpu.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py > $OUT/bench.json 2> $OUT/bench.err || { tail -20 $OUT/bench.err; exit 1; }
cat $OUT/bench.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python bench.py > $OUT/prof_bench.json 2> $OUT/prof.err || { tail -20 $OUT/prof.err; exit 1; }
find $OUT/prof -name '*kernel_stats.csv' | head -1 | xargs -r head -8
