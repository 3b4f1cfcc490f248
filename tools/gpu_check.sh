#!/bin/bash
# tools/gpu_check.sh [tag] — one gpurun call: smoke, GPU parity tests (plain and with poisoned allocations), the bench line under the driver's own invocation and
# under the defaults, rocprofv3 kernel stats of the same command.
# usage: gpurun --timeout 1100 -- 'bash tools/gpu_check.sh [tag]'
set -o pipefail
TAG=${1:-r02}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python3 -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1 || { tail -20 $OUT/smoke.log; exit 1; }
tail -2 $OUT/smoke.log
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; rc=$?
tail -15 $OUT/pytest_gpu.log
[ $rc -eq 0 ] || exit $rc
# once more with every device allocation of the library filled with 0xA5 (VFHIP_DEBUG_POISON): a kernel that reads an intermediate / staging byte
# nothing wrote fails here on every run, not only when the allocator hands back dirty memory
VFHIP_DEBUG_POISON=1 timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $OUT/pytest_gpu_poison.log 2>&1; rc=$?
tail -3 $OUT/pytest_gpu_poison.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err || { tail -20 $OUT/bench_driver.err; exit 1; }
cat $OUT/bench_driver.json
timeout -k 10 300 python3 bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err || { tail -20 $OUT/bench.err; exit 1; }
cat $OUT/bench.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-ceilings > $OUT/prof_bench.json 2> $OUT/prof.err || { tail -20 $OUT/prof.err; exit 1; }
cat $OUT/prof_bench.json
find $OUT/prof -name '*kernel_stats.csv' | head -1 | xargs -r head -8
