"""CPU-side checks of the C-ABI boundary: libvfhip.so loads, exports every symbol include/vfhip.h declares,
and fails loudly (no CPU fallback) when there is no GPU.  No compute calls here."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "vfhip.h")
LIB = os.path.join(ROOT, "gstreamer-metal_amd", "libvfhip.so")


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vfhip_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_expected_entry_points():
    syms = declared_symbols()
    for s in ("vfhip_convertscale_new", "vfhip_convertscale_configure", "vfhip_convertscale_process",
              "vfhip_convertscale_process_device_batch", "vfhip_convertscale_cleanup", "vfhip_convertscale_free",
              "vfhip_device_count", "vfhip_last_error_string"):
        assert s in syms


def test_library_exports_every_declared_symbol():
    assert os.path.exists(LIB), "libvfhip.so not built (run __graft_entry__.build())"
    lib = C.CDLL(LIB)
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, missing
    assert lib.vfhip_abi_version() == 1


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import vfhip
    with pytest.raises(vfhip.VfHipError) as e:
        vfhip.device_count()
    assert e.value.code == -6
    with pytest.raises(vfhip.VfHipError):
        vfhip.ConvertScale(0)


def test_plane_geometry_helpers():
    lib = C.CDLL(LIB)
    NV12, I420, BGRA, UYVY = 2, 3, 0, 4
    assert lib.vfhip_format_n_planes(NV12) == 2 and lib.vfhip_format_n_planes(I420) == 3 and lib.vfhip_format_n_planes(BGRA) == 1
    assert lib.vfhip_plane_width_bytes(NV12, 1, 63) == 64 and lib.vfhip_plane_height(NV12, 1, 35) == 18
    assert lib.vfhip_plane_width_bytes(I420, 2, 63) == 32
    assert lib.vfhip_plane_width_bytes(UYVY, 0, 63) == 128
    assert lib.vfhip_format_n_planes(99) < 0
