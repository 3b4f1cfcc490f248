#!/bin/bash
# quick A/B of bench variants on one box: parity smoke first, then interleaved bench runs per VFHIP_HALF_ROWS
set -o pipefail
TAG=${1:-ab}; OUT=gpurun_out/$TAG; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1 || { tail -5 $OUT/smoke.log; exit 1; }
timeout -k 10 300 python -m pytest tests/test_convertscale_gpu.py -m gpu -x -q -k "half or golden or full_size" > $OUT/pytest.log 2>&1 || { tail -15 $OUT/pytest.log; exit 1; }
tail -1 $OUT/pytest.log
for round in 1 2; do for r in 16 8 4; do
  VFHIP_HALF_ROWS=$r timeout -k 10 120 python bench.py --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('rows=$r', d['value'], d['roofline']['frac'], d['roofline']['kernel_ms'])" | tee -a $OUT/ab.txt
done; done
