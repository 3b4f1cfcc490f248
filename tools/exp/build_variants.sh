#!/bin/bash
# tools/exp/build_variants.sh [-f file] name "flags" [name "flags" ...] — experimental builds of libvfhip.so with extra -D flags on one translation unit
# (default convertscale; -f videofilter etc.) as build/exp/<name>/libvfhip.so, for A/B runs on one GPU box with tools/exp/run_variants*.sh
set -e
cd "$(dirname "$0")/../../gstreamer-metal_amd"
F=convertscale
if [ "$1" = "-f" ]; then F=$2; shift 2; fi
make -s -j8
while [ $# -gt 1 ]; do
  name=$1; flags=$2; shift 2
  mkdir -p build/exp/$name
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -Wno-align-mismatch --offload-arch=gfx950 $flags -c -o build/exp/$name/$F.o csrc/$F.hip
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o build/exp/$name/libvfhip.so $(ls build/obj/*.o | grep -v /$F.o) build/exp/$name/$F.o -lz
  echo "built $name [$flags]"
done
