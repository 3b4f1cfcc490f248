#!/bin/bash
# tools/gpu_final.sh <tag> — the round's closing measurement on ONE box: headline PMC traffic (-> profiles/traffic_latest.json), element PMC passes + VALU
# figures (-> profiles/valu_latest.json), then tools/gpu_check.sh (smoke, GPU tests plain + poisoned, bench under the driver's invocation, rocprofv3 stats).
# Everything keyed by the sources of this tree: run it after the last kernel change.
T=${1:-final}
mkdir -p gpurun_out/$T
bash tools/gpu_pmc.sh ${T}_pmc 512 > gpurun_out/$T/pmc_headline.log 2>&1 || { tail -5 gpurun_out/$T/pmc_headline.log; exit 1; }
bash tools/gpu_pmc_cmd.sh ${T}_pmce tools/pmc_workloads.py > gpurun_out/$T/pmc_elements_digest.txt 2>&1 || { tail -5 gpurun_out/$T/pmc_elements_digest.txt; exit 1; }
make -s -C gstreamer-metal_amd asm > /dev/null 2>&1
python3 tools/valu_roofline.py gpurun_out/${T}_pmce gstreamer-metal_amd/build/*.s > gpurun_out/$T/valu_elements.json
python3 tools/valu_roofline.py gpurun_out/${T}_pmc gstreamer-metal_amd/build/*.s > gpurun_out/$T/valu_headline.json
python3 tools/update_valu.py gpurun_out/$T/valu_elements.json gpurun_out/$T/valu_headline.json > gpurun_out/$T/valu_latest.log
cp profiles/valu_latest.json profiles/traffic_latest.json gpurun_out/$T/
bash tools/gpu_check.sh ${T}_check > gpurun_out/$T/check.log 2>&1; echo "check rc $?" >> gpurun_out/$T/check.log
tail -3 gpurun_out/$T/check.log | cut -c1-300
