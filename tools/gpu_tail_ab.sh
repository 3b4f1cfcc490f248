#!/bin/bash
# A/B of the drain-region size (VFHIP_HALF_TAIL frames in 4-row strips) at the default 128 frames per launch
set -o pipefail
TAG=${1:-tab}; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout -k 10 300 python -m pytest tests/test_convertscale_gpu.py -m gpu -x -q > $OUT/pytest.log 2>&1 || { tail -15 $OUT/pytest.log; exit 1; }
tail -1 $OUT/pytest.log
for round in 1 2; do for t in 0 8 16 24 32; do
  VFHIP_HALF_TAIL=$t timeout -k 10 120 python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('tail=$t', d['value'], d['roofline']['frac'], d['roofline']['kernel_ms'])" | tee -a $OUT/ab.txt
done; done
