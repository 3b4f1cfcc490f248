"""Build-time guard (no GPU needed): no hand-written kernel may spill to scratch memory.

hipcc cross-compiles gfx950 here; `-Rpass-analysis=kernel-resource-usage` reports every kernel's registers and scratch bytes.
A dynamically indexed kernel-argument struct once put k_scale_packed422 into 432 bytes of scratch per lane and made it ten
times slower without changing a single output byte — parity tests cannot see that, this one does."""
import glob
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"
SRCS = sorted(glob.glob(os.path.join(ROOT, "gstreamer-metal_amd", "csrc", "*.hip")))


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
@pytest.mark.parametrize("src", SRCS, ids=[os.path.basename(s) for s in SRCS])
def test_no_kernel_uses_scratch(src):
    r = subprocess.run([HIPCC, "-O3", "-std=c++17", "-ffp-contract=off", "--offload-arch=gfx950", "--cuda-device-only", "-I" + os.path.join(ROOT, "include"),
                        "-I" + os.path.dirname(src), "-c", src, "-Rpass-analysis=kernel-resource-usage", "-o", os.devnull],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    names = re.findall(r"Function Name: (\S+)", r.stderr)
    scratch = [int(v) for v in re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", r.stderr)]
    vgprs = [int(v) for v in re.findall(r"\bVGPRs: (\d+)", r.stderr)]
    assert len(names) == len(scratch) == len(vgprs)
    spilled = {n: s for n, s in zip(names, scratch) if s}
    assert not spilled, f"kernels with scratch memory: {spilled}"
    assert all(v <= 128 for v in vgprs), dict(zip(names, vgprs))      # 128 VGPRs = 4 waves per SIMD: nothing here should need more
