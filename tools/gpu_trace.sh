#!/bin/bash
set -o pipefail
TAG=${1:-tr}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
for f in 128 512; do
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/tr$f -- python bench.py --steps $((2560 / f)) --warmup 3 --frames $f --no-cpu-baseline > $OUT/tr$f.json 2> $OUT/tr$f.err || { tail -5 $OUT/tr$f.err; exit 1; }
cat $OUT/tr$f.json | cut -c1-200
python tools/trace_gaps.py $OUT/tr$f
find $OUT/tr$f -name '*kernel_trace.csv' -size +2M -delete
done
