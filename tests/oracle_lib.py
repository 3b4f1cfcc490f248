"""ctypes loader for the CPU parity oracle (oracle/*.c).  Test infrastructure: only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this."""
import ctypes as C
import json
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "libvfhip_oracle.so")
MATRIX = {"bt601": 0, "bt709": 1, "bt2020": 2}


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


def r4(x):
    return (x + 3) // 4 * 4


def default_colorimetry(h):
    """GStreamer's by-height default when caps carry no colorimetry (SURVEY.md §8c rule 1)."""
    if h >= 2160:
        return "bt2020", "mpeg2"
    if h > 576:
        return "bt709", "mpeg2"
    return "bt601", "jpeg"


def planes(fmt, w, h, raw):
    """Split a GstVideoInfo-default-layout raw frame into (array, stride) planes."""
    raw = np.ascontiguousarray(raw, dtype=np.uint8)
    hp = (h + 1) // 2 * 2
    if fmt == "NV12":
        ys = r4(w)
        return [(raw[: ys * h], ys), (raw[ys * hp: ys * hp + ys * (hp // 2)], ys)]
    if fmt == "I420":
        ys, cs = r4(w), r4((w + 1) // 2)
        uo = ys * hp
        vo = uo + cs * (hp // 2)
        return [(raw[: ys * h], ys), (raw[uo: uo + cs * (hp // 2)], cs), (raw[vo: vo + cs * (hp // 2)], cs)]
    if fmt in ("BGRA", "RGBA"):
        return [(raw[: 4 * w * h], 4 * w)]
    raise ValueError(fmt)


class Oracle:
    def __init__(self, lib):
        self.lib = lib
        lib.gst114_convertscale_yuv420.restype = C.c_int

    @staticmethod
    def _p(a, off=0):
        return C.c_void_p(a.ctypes.data + off)

    def convertscale(self, fmt, w, h, raw, colorimetry, chroma_site, method, out_format, ow, oh):
        """raw: bytes/array in GstVideoInfo default layout. Returns (oh, ow, 4) uint8."""
        if colorimetry is None:
            colorimetry, chroma_site = default_colorimetry(h)
        pl = planes(fmt, w, h, np.frombuffer(raw, np.uint8) if isinstance(raw, (bytes, bytearray)) else raw)
        out = np.zeros((oh, ow, 4), np.uint8)
        cos = 1 if chroma_site == "mpeg2" else 0
        meth = 1 if method == "nearest" else 0
        ofmt = 1 if out_format == "RGBA" else 0
        if fmt == "NV12":
            (y, ys), (uv, us) = pl
            rc = self.lib.gst114_convertscale_yuv420(self._p(y), ys, self._p(uv), us, self._p(uv, 1), us, 0, w, h,
                                                     MATRIX[colorimetry], cos, ofmt, meth, self._p(out), ow * 4, ow, oh)
        elif fmt == "I420":
            (y, ys), (u, us), (v, vs) = pl
            rc = self.lib.gst114_convertscale_yuv420(self._p(y), ys, self._p(u), us, self._p(v), vs, 1, w, h,
                                                     MATRIX[colorimetry], cos, ofmt, meth, self._p(out), ow * 4, ow, oh)
        else:
            raise ValueError(fmt)
        if rc != 0:
            raise RuntimeError(f"oracle rc={rc}")
        return out


def load():
    build()
    return Oracle(C.CDLL(LIB))


def load_golden(name="convertscale_gst114.npz"):
    z = np.load(os.path.join(ROOT, "tests", "golden", name), allow_pickle=False)
    manifest = json.loads(bytes(z["manifest"]).decode())
    return manifest, z
