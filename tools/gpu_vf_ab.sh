#!/bin/bash
set -o pipefail
TAG=${1:-vfab}; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout -k 10 300 python -m pytest tests/test_metal_elements_gpu.py -m gpu -x -q -k videofilter > $OUT/pytest.log 2>&1 || { tail -15 $OUT/pytest.log; exit 1; }
tail -1 $OUT/pytest.log
for v in 0 1 2 3; do
  echo "variant $v" | tee -a $OUT/ab.txt
  VFHIP_VF_TILE=$v timeout -k 10 120 python tools/bench_vf_ablation.py 2>/dev/null | grep -E "sharp|all-15" | tee -a $OUT/ab.txt
done
