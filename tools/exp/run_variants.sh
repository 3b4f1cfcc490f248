#!/bin/bash
# tools/exp/run_variants.sh [rounds] — on the GPU box: for every build/exp/<name>/libvfhip.so, smoke (bit-exact vs the oracle) and the
# headline bench, `rounds` interleaved rounds; prints frames/s per variant.  The variant is selected with $VFHIP_LIB (read by vfhip.py);
# the product library is never overwritten.
set -o pipefail
R=${1:-2}
cd "$(dirname "$0")/../.."
for r in $(seq 1 $R); do
  for d in gstreamer-metal_amd/build/exp/*/; do
    n=$(basename $d)
    [ -f $d/libvfhip.so ] || continue
    export VFHIP_LIB=$PWD/$d/libvfhip.so
    if [ $r -eq 1 ]; then python3 -c "import __graft_entry__ as g; g.smoke()" > /tmp/smoke_$n.log 2>&1 || { echo "$n: SMOKE FAILED"; tail -3 /tmp/smoke_$n.log; continue; }; fi
    python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-ceilings --no-others 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$n round $r: %.0f frames/s  frac %.4f  kernel_ms %.4f  sclk %s' % (d['value'], d['roofline']['frac'], d['roofline']['kernel_ms'], d['clocks'].get('sclk_MHz')))"
  done
done
