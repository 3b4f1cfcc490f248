#!/bin/bash
# tools/gpu_fuzz.sh <tag> <seconds> [first seed] [metal every n-th run, default 5] — a fuzz campaign on the GPU box: five workers (the box allows six GPU processes) take turns with
# tools/fuzz_gst_exact.py (3000 cases a run, 4 of 5 runs) and tools/fuzz_metal.py (1000 cases), fresh seeds, until <seconds> have passed; every device
# allocation poisoned, the oracle single-threaded (or $VFHIP_ORACLE_THREADS: on these small frames 16 threads halve the case rate).    One line per run in gpurun_out/<tag>/summary.txt; mismatch records land in gpurun_out/fuzz_dumps/.
T=$1; SECS=$2; SEED=${3:-7000}; EVERY=${4:-5}
OUT=gpurun_out/$T; mkdir -p $OUT
export VFHIP_DEBUG_POISON=1 VFHIP_ORACLE_THREADS=${VFHIP_ORACLE_THREADS:-1}
END=$(( $(date +%s) + SECS ))
worker () {
  local w=$1 s=$(( SEED + $1 * 1000 ))
  while [ $(date +%s) -lt $END ]; do
    if [ $(( s % EVERY )) -eq $(( EVERY - 1 )) ]; then python3 tools/fuzz_metal.py 1000 $s > $OUT/metal_$s.log 2>&1; echo "metal $s rc $? $(tail -1 $OUT/metal_$s.log)" >> $OUT/summary.txt
    else python3 tools/fuzz_gst_exact.py 3000 $s > $OUT/gst_$s.log 2>&1; echo "gst $s rc $? $(tail -1 $OUT/gst_$s.log)" >> $OUT/summary.txt; fi
    s=$(( s + 1 ))
  done
}
for w in 0 1 2 3 4; do worker $w & done
wait
sort $OUT/summary.txt | awk '{print $1}' | uniq -c
grep -v " rc 0 " $OUT/summary.txt | head
ls gpurun_out/fuzz_dumps 2>/dev/null | head
