#!/bin/bash
# tools/exp/run_variants_cmd.sh rounds 'command' — on the GPU box: run `command` with every build/exp/<name>/libvfhip.so selected through
# $VFHIP_LIB (read by vfhip.py), `rounds` interleaved rounds (A/B on one box).  The product library is never overwritten.
R=$1; shift
cd "$(dirname "$0")/../.."
for r in $(seq 1 $R); do
  for d in gstreamer-metal_amd/build/exp/*/; do
    n=$(basename $d); [ -f $d/libvfhip.so ] || continue
    echo "== $n round $r"; VFHIP_LIB=$PWD/$d/libvfhip.so bash -c "$*" 2>&1 | grep -v amdgpu.ids
  done
done
