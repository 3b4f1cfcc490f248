#!/bin/bash
set -o pipefail
TAG=${1:-ep}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python tools/bench_vf_ablation.py > $OUT/abl.jsonl 2> $OUT/abl.err || { tail -5 $OUT/abl.err; exit 1; }
python tools/trace_gaps.py $OUT/prof k_vf
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof2 -- python tools/bench_elements.py main > $OUT/elem.jsonl 2> $OUT/elem.err || { tail -5 $OUT/elem.err; exit 1; }
find $OUT/prof2 -name '*kernel_stats.csv' | head -1 | xargs -r cut -c1-150
find $OUT -name '*kernel_trace.csv' -size +3M -delete
