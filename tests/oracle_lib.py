"""ctypes loader for the CPU parity oracle (oracle/*.c).  Test infrastructure: only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this."""
import ctypes as C
import json
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.environ.get("VFHIP_ORACLE_LIB") or os.path.join(ORACLE_DIR, "libvfhip_oracle.so")   # override: the sanitizer build (tests/test_parsers_asan.py)
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
    if fmt in ("UYVY", "YUY2"):
        return [(raw[: r4(2 * w) * h], r4(2 * w))]
    raise ValueError(fmt)


class Oracle:
    def __init__(self, lib):
        self.lib = lib
        lib.gst114_convertscale_yuv420.restype = C.c_int

    @staticmethod
    def _p(a, off=0):
        return C.c_void_p(a.ctypes.data + off)

    def _yuv_spec(self, fmt, w, h, arr):
        """sample addressing of a YUV frame in GstVideoInfo default layout for gst114_yuv_to_yuv: (y, ys, ystep, u, v, cs, cstep, is420)"""
        lay, _ = raw_layout(fmt, w, h)
        if fmt == "NV12":
            return self._p(arr, lay[0][0]), lay[0][1], 1, self._p(arr, lay[1][0]), self._p(arr, lay[1][0] + 1), lay[1][1], 2, 1
        if fmt == "I420":
            return self._p(arr, lay[0][0]), lay[0][1], 1, self._p(arr, lay[1][0]), self._p(arr, lay[2][0]), lay[1][1], 1, 1
        yo, uo, vo = (0, 1, 3) if fmt == "YUY2" else (1, 0, 2)
        return self._p(arr, yo), lay[0][1], 2, self._p(arr, uo), self._p(arr, vo), lay[0][1], 4, 0

    def yuv_to_yuv(self, fmt, w, h, raw, mat, cos, out_format, mat_out, cos_out):
        """videoconvert YUV -> YUV at one size with a matrix and / or siting change (gst114_yuv_to_yuv): the raw output frame"""
        raw = np.ascontiguousarray(raw, np.uint8)
        out = np.zeros(raw_layout(out_format, w, h)[1], np.uint8)
        i, o = self._yuv_spec(fmt, w, h, raw), self._yuv_spec(out_format, w, h, out)
        rc = self.lib.gst114_yuv_to_yuv(i[0], i[1], i[2], i[3], i[4], i[5], i[6], i[7], w, h, mat, cos, mat_out, cos_out,
                                        o[0], o[1], o[2], o[3], o[4], o[5], o[6], o[7])
        if rc != 0:
            raise RuntimeError(f"oracle rc={rc}")
        return out

    def convertscale(self, fmt, w, h, raw, colorimetry, chroma_site, method, out_format, ow, oh, out_chroma_site=None, out_colorimetry=None):
        """GStreamer 1.14 `videoconvert ! videoscale` on one frame in GstVideoInfo default layout, any of
        {NV12, I420, BGRA, RGBA, UYVY, YUY2} -> any of them.  RGB outputs return (oh, ow, 4); YUV outputs the raw output frame."""
        if colorimetry is None:
            colorimetry, chroma_site = default_colorimetry(h)
        raw = np.ascontiguousarray(np.frombuffer(raw, np.uint8) if isinstance(raw, (bytes, bytearray)) else raw, dtype=np.uint8)
        pl = planes(fmt, w, h, raw)
        cos = 1 if chroma_site == "mpeg2" else 0
        cos_out = cos if out_chroma_site is None else (1 if out_chroma_site == "mpeg2" else 0)   # only the NV12 <-> packed and RGB -> YUV steps look at it
        meth = {"bilinear": 0, "nearest": 1, "bicubic": 2}[method]
        mat = MATRIX[colorimetry]
        L = self.lib
        yuv_in, yuv_out = fmt in ("NV12", "I420"), out_format in ("NV12", "I420")
        if out_format in ("BGRA", "RGBA"):
            out = np.zeros((oh, ow, 4), np.uint8)
            ofmt = 1 if out_format == "RGBA" else 0
            if fmt == "NV12":
                (y, ys), (uv, us) = pl
                rc = L.gst114_convertscale_yuv420(self._p(y), ys, self._p(uv), us, self._p(uv, 1), us, 0, w, h, mat, cos, ofmt, meth,
                                                  self._p(out), ow * 4, ow, oh)
            elif fmt == "I420":
                (y, ys), (u, us), (v, vs) = pl
                rc = L.gst114_convertscale_yuv420(self._p(y), ys, self._p(u), us, self._p(v), vs, 1, w, h, mat, cos, ofmt, meth,
                                                  self._p(out), ow * 4, ow, oh)
            elif fmt in ("UYVY", "YUY2"):
                stride = r4(2 * w)
                rc = L.gst114_convertscale_packed422(self._p(raw), stride, int(fmt == "YUY2"), w, h, mat, cos, ofmt, meth, self._p(out), ow * 4, ow, oh)
            else:
                src = pl[0][0].reshape(h, w, 4)
                if fmt != out_format:
                    src = np.ascontiguousarray(src[..., [2, 1, 0, 3]])
                rc = L.gst114_scale_4u8(self._p(src), 4 * w, w, h, self._p(out), 4 * ow, ow, oh, meth)
            if rc != 0:
                raise RuntimeError(f"oracle rc={rc}")
            return out
        near = method == "nearest"
        sfx = {"bilinear": "", "nearest": "_nearest", "bicubic": "_cubic"}[method]
        # YUV -> YUV with a matrix change (any formats), or NV12 <-> I420 with a siting change: videoconvert's generic path at the
        # input size (gst114_yuv_to_yuv), then the usual videoscale stage on the converted frame
        mat_out = mat if out_colorimetry is None else MATRIX[out_colorimetry]
        any_yuv_in = fmt in ("NV12", "I420", "UYVY", "YUY2")
        if any_yuv_in and (mat_out != mat or (yuv_in and yuv_out and fmt != out_format and cos != cos_out)):
            mid = self.yuv_to_yuv(fmt, w, h, raw, mat, cos, out_format, mat_out, cos_out)
            if (ow, oh) == (w, h):
                return mid
            return self.convertscale(out_format, w, h, mid, {v: k for k, v in MATRIX.items()}[mat_out], "mpeg2" if cos_out else "jpeg", method, out_format, ow, oh)
        if out_format in ("UYVY", "YUY2"):
            # videoconvert at the input size -> packed frame of the output format, then videoscale on the packed frame
            ms, yuy2 = r4(2 * w), int(out_format == "YUY2")
            mid = np.zeros(ms * h, np.uint8)
            if fmt == "NV12":
                (y, ys), (uv, us) = pl
                rc = L.gst114_yuv420_to_packed422(self._p(y), ys, self._p(uv), us, self._p(uv, 1), us, 0, w, h, cos, cos_out, yuy2, self._p(mid), ms)
            elif fmt == "I420":
                (y, ys), (u, us), (v, vs) = pl
                rc = L.gst114_yuv420_to_packed422(self._p(y), ys, self._p(u), us, self._p(v), vs, 1, w, h, cos, cos, yuy2, self._p(mid), ms)
            elif fmt in ("UYVY", "YUY2"):
                rc = L.gst114_packed422_swizzle(self._p(raw), ms, int(fmt == "YUY2"), w, h, yuy2, self._p(mid), ms)
            else:
                rc = L.gst114_rgb_to_packed422(self._p(pl[0][0]), 4 * w, 1 if fmt == "RGBA" else 0, w, h, mat, cos, yuy2, self._p(mid), ms)
            assert rc == 0
            os_ = r4(2 * ow)
            out = np.zeros(os_ * oh, np.uint8)
            rc = getattr(L, "gst114_scale_packed422" + sfx)(self._p(mid), ms, yuy2, w, h, self._p(out), os_, ow, oh)
            if rc != 0:
                raise RuntimeError(f"oracle rc={rc}")
            return out
        # stage 1: videoconvert at the input size -> planes of the output format
        cw, ch = (w + 1) // 2, (h + 1) // 2
        Y = np.zeros((h, w), np.uint8)
        if out_format == "NV12":
            C = [np.zeros((ch, 2 * cw), np.uint8)]
        else:
            C = [np.zeros((ch, cw), np.uint8), np.zeros((ch, cw), np.uint8)]
        if fmt in ("UYVY", "YUY2"):
            planar = out_format == "I420"
            up = C[0]
            vp = C[1] if planar else C[0]
            rc = L.gst114_packed422_to_yuv420(self._p(raw), r4(2 * w), int(fmt == "YUY2"), w, h, cos, cos_out, int(planar), self._p(Y), w,
                                              self._p(up), up.strides[0], self._p(vp, 0 if planar else 1), vp.strides[0])
            assert rc == 0
        elif not yuv_in:
            src = pl[0][0]
            planar = out_format == "I420"
            up = C[0]
            vp = C[1] if planar else C[0]
            rc = L.gst114_rgb_to_yuv420(self._p(src), 4 * w, 1 if fmt == "RGBA" else 0, w, h, mat, cos, int(planar), self._p(Y), w,
                                        self._p(up), up.strides[0], self._p(vp), vp.strides[0])
            assert rc == 0
        else:
            Y[:] = pl[0][0].reshape(-1, pl[0][1])[:h, :w]
            if fmt == "NV12":
                uv = pl[1][0].reshape(-1, pl[1][1])[:ch, :2 * cw]
                u, v = uv[:, 0::2], uv[:, 1::2]
            else:
                u = pl[1][0].reshape(-1, pl[1][1])[:ch, :cw]
                v = pl[2][0].reshape(-1, pl[2][1])[:ch, :cw]
            if out_format == "NV12":
                C[0][:, 0::2], C[0][:, 1::2] = u, v
            else:
                C[0][:], C[1][:] = u, v
        # stage 2: videoscale, plane by plane
        ocw, och = (ow + 1) // 2, (oh + 1) // 2
        lay, size = raw_layout(out_format, ow, oh)
        out = np.zeros(size, np.uint8)

        def scale(src, sw, sh, n, off, stride, dw, dh, chroma=False):
            src = np.ascontiguousarray(src)
            args = (self._p(src), src.strides[0], sw, sh, n, self._p(out, off), stride, dw, dh)
            # videoscale method=catrom: catrom on the luma plane, GstVideoConverter's un-limited LINEAR taps on the chroma planes
            rc = L.gst114_scale_plane_cubic(*args, int(chroma)) if method == "bicubic" else getattr(L, "gst114_scale_plane" + sfx)(*args)
            if rc != 0:
                raise RuntimeError(f"oracle rc={rc}")
        scale(Y, w, h, 1, lay[0][0], lay[0][1], ow, oh)
        if out_format == "NV12":
            scale(C[0], cw, ch, 2, lay[1][0], lay[1][1], ocw, och, chroma=True)
        else:
            scale(C[0], cw, ch, 1, lay[1][0], lay[1][1], ocw, och, chroma=True)
            scale(C[1], cw, ch, 1, lay[2][0], lay[2][1], ocw, och, chroma=True)
        return out


def load():
    """The checker runs single-threaded unless VFHIP_ORACLE_THREADS says otherwise: gst114.c's OpenMP loops are there for bench.py's CPU baseline
    (which sets its own thread count).  Three one-off differences in ~80,000 GPU-box fuzz cases turned out to be the ORACLE's side (its second run
    agreed with the library, tools/fuzz_gst_exact.py re-runs both); no race was found — 100,000 repeats with 8 and 64 threads and an ASan run are
    clean — but a checker has no use for threads."""
    build()
    lib = C.CDLL(LIB)
    lib.gst114_set_threads(int(os.environ.get("VFHIP_ORACLE_THREADS", "1")))
    return Oracle(lib)


def load_golden(name="convertscale_gst114.npz"):
    z = np.load(os.path.join(ROOT, "tests", "golden", name), allow_pickle=False)
    manifest = json.loads(bytes(z["manifest"]).decode())
    return manifest, z


# ---- metalref (oracle/metalref.c): float restatement of the reference's Metal shaders ---------------------------
FMT = {"BGRA": 0, "RGBA": 1, "NV12": 2, "I420": 3, "UYVY": 4, "YUY2": 5}


class MrImg(C.Structure):
    _fields_ = [("p", C.c_void_p * 3), ("s", C.c_int32 * 3), ("w", C.c_int32), ("h", C.c_int32), ("fmt", C.c_int32),
                ("m709", C.c_int32)]


class MrFilterParams(C.Structure):
    _fields_ = [("brightness", C.c_float), ("contrast", C.c_float), ("saturation", C.c_float), ("hue", C.c_float),
                ("gamma", C.c_float), ("sharpness", C.c_float), ("sepia", C.c_float), ("noise", C.c_float),
                ("vignette", C.c_float), ("invert", C.c_int32), ("chroma_key_enabled", C.c_int32), ("key_r", C.c_float),
                ("key_g", C.c_float), ("key_b", C.c_float), ("key_tolerance", C.c_float), ("key_smoothness", C.c_float),
                ("frame_index", C.c_uint32)]


class MrPad(C.Structure):
    _fields_ = [("img", MrImg), ("xpos", C.c_int32), ("ypos", C.c_int32), ("width", C.c_int32), ("height", C.c_int32),
                ("alpha", C.c_double), ("blend", C.c_int32)]


def raw_layout(fmt, w, h):
    """GstVideoInfo default layout: [(offset, stride)], size."""
    hp = (h + 1) // 2 * 2
    if fmt in ("BGRA", "RGBA"):
        return [(0, 4 * w)], 4 * w * h
    if fmt in ("UYVY", "YUY2"):
        s = r4(2 * w)
        return [(0, s)], s * h
    if fmt == "NV12":
        s = r4(w)
        return [(0, s), (s * hp, s)], s * hp + s * (hp // 2)
    s, cs = r4(w), r4((w + 1) // 2)
    uo = s * hp
    vo = uo + cs * (hp // 2)
    return [(0, s), (uo, cs), (vo, cs)], vo + cs * (hp // 2)


def mr_img(fmt, w, h, raw, m709=False):
    im = MrImg()
    pl, size = raw_layout(fmt, w, h)
    assert raw.size >= size and raw.dtype == np.uint8 and raw.flags["C_CONTIGUOUS"]
    for i, (off, stride) in enumerate(pl):
        im.p[i] = raw.ctypes.data + off
        im.s[i] = stride
    im.w, im.h, im.fmt, im.m709 = w, h, FMT[fmt], int(m709)
    return im


class MetalRef:
    def __init__(self, lib):
        self.lib = lib

    def convertscale(self, in_fmt, w, h, raw, out_fmt, ow, oh, linear=True, add_borders=False, border=0xFF000000,
                     m709_in=False, m709_out=False):
        raw = np.ascontiguousarray(raw, np.uint8)
        out = np.zeros(raw_layout(out_fmt, ow, oh)[1], np.uint8)
        i, o = mr_img(in_fmt, w, h, raw, m709_in), mr_img(out_fmt, ow, oh, out, m709_out)
        assert self.lib.metalref_convertscale(C.byref(i), C.byref(o), int(linear), int(add_borders), C.c_uint32(border)) == 0
        return out

    def deinterlace(self, fmt, w, h, cur, prev, method, tff=True, threshold=0.1, m709=False):
        cur = np.ascontiguousarray(cur, np.uint8)
        out = np.zeros(raw_layout(fmt, w, h)[1], np.uint8)
        c, o = mr_img(fmt, w, h, cur, m709), mr_img(fmt, w, h, out, m709)
        pp = None
        if prev is not None:
            prev = np.ascontiguousarray(prev, np.uint8)
            pimg = mr_img(fmt, w, h, prev, m709)
            pp = C.byref(pimg)
        assert self.lib.metalref_deinterlace(C.byref(c), pp, C.byref(o), method, int(tff), C.c_float(threshold)) == 0
        return out

    def videofilter(self, in_fmt, w, h, raw, out_fmt, params, lut=None, m709=False):
        raw = np.ascontiguousarray(raw, np.uint8)
        out = np.zeros(raw_layout(out_fmt, w, h)[1], np.uint8)
        i, o = mr_img(in_fmt, w, h, raw, m709), mr_img(out_fmt, w, h, out, m709)
        lp, n = None, 0
        if lut is not None:
            lut = np.ascontiguousarray(lut, np.float32)
            n = int(round((lut.size // 4) ** (1 / 3)))
            lp = lut.ctypes.data_as(C.c_void_p)
        assert self.lib.metalref_videofilter(C.byref(i), C.byref(o), C.byref(params), lp, n) == 0
        return out

    def compositor(self, out_fmt, ow, oh, pads, background, m709_out=False):
        """pads: list of (fmt, w, h, raw, xpos, ypos, width, height, alpha, blend_int[, m709])."""
        out = np.zeros(raw_layout(out_fmt, ow, oh)[1], np.uint8)
        o = mr_img(out_fmt, ow, oh, out, m709_out)
        keep = [np.ascontiguousarray(p[3], np.uint8) for p in pads]
        arr = (MrPad * max(len(pads), 1))()
        for k, p in enumerate(pads):
            arr[k].img = mr_img(p[0], p[1], p[2], keep[k], p[10] if len(p) > 10 else False)
            arr[k].xpos, arr[k].ypos, arr[k].width, arr[k].height, arr[k].alpha, arr[k].blend = p[4], p[5], p[6], p[7], p[8], p[9]
        assert self.lib.metalref_compositor(arr, len(pads), background, C.byref(o)) == 0
        return out


def _mr_transform(self, in_fmt, w, h, raw, out_fmt, method, crop=(0, 0, 0, 0), m709=False):
    raw = np.ascontiguousarray(raw, np.uint8)
    out = np.zeros(raw_layout(out_fmt, w, h)[1], np.uint8)
    i, o = mr_img(in_fmt, w, h, raw, m709), mr_img(out_fmt, w, h, out, m709)
    assert self.lib.metalref_transform(C.byref(i), C.byref(o), method, *crop) == 0
    return out


MetalRef.transform = _mr_transform


def _mr_overlay(self, in_fmt, w, h, raw, out_fmt, image, x=0.0, y=0.0, width=0.0, height=0.0, alpha=1.0, m709=False):
    """image: (ih, iw, 4) uint8 as the shader sees it, or None"""
    raw = np.ascontiguousarray(raw, np.uint8)
    out = np.zeros(raw_layout(out_fmt, w, h)[1], np.uint8)
    i, o = mr_img(in_fmt, w, h, raw, m709), mr_img(out_fmt, w, h, out, m709)
    ov = None
    if image is not None:
        image = np.ascontiguousarray(image, np.uint8)
        ih, iw = image.shape[:2]
        ovi = mr_img("RGBA", iw, ih, image.reshape(-1))
        ov = C.byref(ovi)
        width, height = (width if width > 0 else iw), (height if height > 0 else ih)
    f = C.c_float
    assert self.lib.metalref_overlay(C.byref(i), C.byref(o), ov, f(x), f(y), f(width), f(height), f(alpha)) == 0
    return out


MetalRef.overlay = _mr_overlay


def load_metalref():
    build()
    lib = C.CDLL(LIB)
    for n in ("metalref_convertscale", "metalref_deinterlace", "metalref_videofilter", "metalref_compositor", "metalref_transform", "metalref_overlay"):
        getattr(lib, n).restype = C.c_int
    return MetalRef(lib)


def mr_filter_params(p):
    """vfhip.VideoFilterParams -> MrFilterParams (same field order)."""
    return MrFilterParams(*[getattr(p, f[0]) for f in p._fields_])


def letterbox_rect(w, h, ow, oh):
    """the reference's centred aspect-preserving rectangle (metalconvertscalerenderer.m:137-166; float32 ratios, round to nearest)"""
    import math
    src, dst = np.float32(w) / np.float32(h), np.float32(ow) / np.float32(oh)
    rw, rh = ow, oh
    if src > dst:
        rh = int(math.floor(float(oh) * float(np.float32(dst / src)) + 0.5))
    else:
        rw = int(math.floor(float(ow) * float(np.float32(src / dst)) + 0.5))
    rw, rh = min(max(rw, 1), ow), min(max(rh, 1), oh)
    return (ow - rw) // 2, (oh - rh) // 2, rw, rh


def convertscale_with_borders(oracle, fmt, w, h, raw, colorimetry, chroma_site, method, out_format, ow, oh, border_argb):
    """add-borders on top of the gst-exact path: the rectangle holds what `videoconvert ! videoscale` gives at
    its size, the rest is the border colour (RGB outputs: in the output's byte order; YUV outputs: through the RGB -> YUV matrix).
    Returns the raw output frame, or None when a YUV output's rectangle is not on chroma-sample boundaries (metal arithmetic then)."""
    rx, ry, rw, rh = letterbox_rect(w, h, ow, oh)
    a, r, g, b = (border_argb >> 24) & 0xff, (border_argb >> 16) & 0xff, (border_argb >> 8) & 0xff, border_argb & 0xff
    inner = np.asarray(oracle.convertscale(fmt, w, h, raw, colorimetry, chroma_site, method, out_format, rw, rh))
    if out_format in ("BGRA", "RGBA"):
        out = np.empty((oh, ow, 4), np.uint8)
        out[:] = np.array([r, g, b, a] if out_format == "RGBA" else [b, g, r, a], np.uint8)
        out[ry:ry + rh, rx:rx + rw] = inner
        return out.reshape(-1)
    packed = out_format in ("UYVY", "YUY2")
    if rw == ow and rh == oh:
        return inner.reshape(-1)                          # same aspect ratio: no borders
    if ((rx | rw) & 1) or (not packed and ((ry | rh) & 1)):
        return None
    m = {"bt601": (66, 129, 25, -38, -74, 112, 112, -94, -18), "bt709": (47, 157, 16, -26, -87, 112, 112, -102, -10),
         "bt2020": (58, 149, 13, -31, -81, 112, 112, -103, -9)}[colorimetry]
    Y, U, V = ((m[0] * r + m[1] * g + m[2] * b) >> 8) + 16, ((m[3] * r + m[4] * g + m[5] * b) >> 8) + 128, ((m[6] * r + m[7] * g + m[8] * b) >> 8) + 128
    opl, osz = raw_layout(out_format, ow, oh)
    ipl, _ = raw_layout(out_format, rw, rh)
    want = np.zeros(osz, np.uint8)
    if packed:
        (o0, os_), (i0, is_) = opl[0], ipl[0]
        rows = want[o0: o0 + os_ * oh].reshape(oh, os_)
        rows[:, : 4 * ((ow + 1) // 2)] = np.tile(np.array([Y, U, Y, V] if out_format == "YUY2" else [U, Y, V, Y], np.uint8), (ow + 1) // 2)
        rows[ry: ry + rh, 2 * rx: 2 * rx + 2 * rw] = inner[i0: i0 + is_ * rh].reshape(rh, is_)[:, : 2 * rw]
        return want
    for k, ((oo, os_), (io, is_)) in enumerate(zip(opl, ipl)):
        sub, n = (1 if k == 0 else 2), (2 if (out_format == "NV12" and k == 1) else 1)
        pw, ph = (ow if k == 0 else (ow + 1) // 2), (oh if k == 0 else (oh + 1) // 2)
        irows = rh if k == 0 else (rh + 1) // 2
        plane = want[oo: oo + os_ * ph].reshape(ph, os_)
        if k == 0:
            plane[:, :pw] = Y
        elif out_format == "NV12":
            plane[:, 0: 2 * pw: 2], plane[:, 1: 2 * pw: 2] = U, V
        else:
            plane[:, :pw] = U if k == 1 else V
        iw_ = rw if k == 0 else (rw + 1) // 2
        plane[ry // sub: ry // sub + irows, n * (rx // sub): n * (rx // sub) + n * iw_] = inner[io: io + is_ * irows].reshape(irows, is_)[:, : n * iw_]
    return want
