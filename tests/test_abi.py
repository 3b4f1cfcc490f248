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


def test_header_is_plain_c_and_a_c_client_links(tmp_path):
    """include/vfhip.h compiles as C99 (pedantic), and a C program written the way a reference maintainer would call
    the boundary links against libvfhip.so and runs its GPU-free part"""
    import shutil
    import subprocess
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    src = tmp_path / "client.c"
    src.write_text(r'''
#include <stdio.h>
#include <string.h>
#include "vfhip.h"
int main (void)
{
  VfHipVideoInfo in, out;
  VfHipFrame f;
  VfHipVideoFilterParams vp;
  VfHipOverlayParams op;
  memset (&in, 0, sizeof in); memset (&out, 0, sizeof out); memset (&f, 0, sizeof f); memset (&vp, 0, sizeof vp); memset (&op, 0, sizeof op);
  in.format = VFHIP_FORMAT_NV12; in.width = 3840; in.height = 2160; in.color_matrix = VFHIP_MATRIX_BT2020; in.chroma_site = VFHIP_CHROMA_SITE_H_COSITED;
  out.format = VFHIP_FORMAT_BGRA; out.width = 1920; out.height = 1080;
  if (vfhip_abi_version () != VFHIP_ABI_VERSION) return 2;
  if (vfhip_plane_width_bytes (in.format, 1, in.width) != 3840 || vfhip_plane_height (in.format, 1, in.height) != 1080) return 3;
  if (vfhip_convertscale_configure (NULL, &in, &out, VFHIP_SCALE_BILINEAR, 0, 0xFF000000u, VFHIP_NUMERICS_GST_EXACT) != VFHIP_ERR_INVALID) return 4;
  if (!strstr (vfhip_last_error_string (), "null")) return 5;
  printf ("devices: %d\n", vfhip_device_count ());     /* negative status without a GPU: no CPU fallback */
  return 0;
}
''')
    exe = tmp_path / "client"
    libdir = os.path.dirname(LIB)
    r = subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                        "-L", libdir, "-lvfhip", f"-Wl,-rpath,{libdir}"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert "devices:" in r.stdout
