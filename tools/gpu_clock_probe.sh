#!/bin/bash
# sample sclk / power with rocm-smi while the headline kernel runs back to back for ~8 s
OUT=gpurun_out/${1:-clk}; mkdir -p $OUT
timeout -k 10 120 python tools/clock_series.py 8 128 > $OUT/series.txt 2>&1 &
PID=$!
sleep 3.5
for i in 1 2 3 4 5 6; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power \(W\)|Socket" ; sleep 0.6; done | tee $OUT/smi.txt
wait $PID
tail -3 $OUT/series.txt
