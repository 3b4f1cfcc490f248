#!/bin/bash
# tools/gpu_vf.sh <tag> — on the GPU box: the video filter's parity tests, the fast path against the oracle stage by stage (three modes), the
# ablation table and the C3 lines of bench_elements.py.  Output under gpurun_out/<tag>/.
T=${1:-vf}; O=gpurun_out/$T; mkdir -p $O
python -m pytest tests/test_metal_elements_gpu.py -q -x -k "videofilter" > $O/pytest_vf.txt 2>&1; echo "pytest rc $?" >> $O/pytest_vf.txt
python tools/exp/vf_fast_vs_oracle.py > $O/fast_vs_oracle.jsonl 2>$O/fast.err
VFHIP_VF_LUT32=1 python tools/exp/vf_fast_vs_oracle.py >> $O/fast_vs_oracle.jsonl 2>>$O/fast.err
VFHIP_VF_EXACT=1 python tools/exp/vf_fast_vs_oracle.py >> $O/fast_vs_oracle.jsonl 2>>$O/fast.err
python tools/bench_vf_ablation.py > $O/vf_ablation.jsonl 2>&1
VFHIP_VF_EXACT=1 python tools/bench_vf_ablation.py > $O/vf_ablation_exact.jsonl 2>&1
VFHIP_VF_LUT32=1 python tools/bench_vf_ablation.py > $O/vf_ablation_lut32.jsonl 2>&1
tail -3 $O/pytest_vf.txt
