#!/bin/bash
# A/B of frames-per-launch (launch tail effect) on one box
set -o pipefail
TAG=${1:-fab}; OUT=gpurun_out/$TAG; mkdir -p $OUT
for round in 1 2; do for f in 128 512 32; do
  timeout -k 10 120 python bench.py --steps $((5120 / f)) --warmup 3 --frames $f --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('frames=$f', d['value'], d['roofline']['frac'], d['roofline']['kernel_ms'])" | tee -a $OUT/ab.txt
done; done
