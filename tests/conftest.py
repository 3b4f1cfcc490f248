import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "gstreamer-metal_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # a checkout without built artefacts (they are git-ignored): build what the tests load, like __graft_entry__.build()
    import subprocess
    lib = os.path.join(ROOT, "gstreamer-metal_amd", "libvfhip.so")
    if not os.path.exists(lib) and os.path.exists("/opt/rocm/bin/hipcc"):
        subprocess.call(["make", "-s", "-C", os.path.join(ROOT, "gstreamer-metal_amd")])
    plugin = os.path.join(ROOT, "gstreamer-metal_amd", "gst", "libgstvfhip.so")
    if os.path.exists(lib) and not os.path.exists(plugin) and os.path.exists("/opt/conda/include/gstreamer-1.0"):
        subprocess.call(["make", "-s", "-C", os.path.join(ROOT, "gstreamer-metal_amd", "gst")])


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    return oracle_lib.load()


@pytest.fixture(scope="session")
def vfhip():
    import vfhip as m
    return m


@pytest.fixture(scope="session")
def metalref():
    import oracle_lib
    return oracle_lib.load_metalref()
