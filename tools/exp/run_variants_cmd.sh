#!/bin/bash
# tools/exp/run_variants_cmd.sh rounds 'command' — on the GPU box: run `command` with every build/exp/<name>/libvfhip.so in place of
# the product library, `rounds` interleaved rounds (A/B on one box)
R=$1; shift
cd "$(dirname "$0")/../.."
cp gstreamer-metal_amd/libvfhip.so /tmp/libvfhip_orig.so
for r in $(seq 1 $R); do
  for d in gstreamer-metal_amd/build/exp/*/; do
    n=$(basename $d); [ -f $d/libvfhip.so ] || continue
    cp $d/libvfhip.so gstreamer-metal_amd/libvfhip.so
    echo "== $n round $r"; bash -c "$*" 2>&1 | grep -v amdgpu.ids
  done
done
cp /tmp/libvfhip_orig.so gstreamer-metal_amd/libvfhip.so
