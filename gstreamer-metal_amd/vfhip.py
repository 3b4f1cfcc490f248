"""vfhip.py — thin ctypes binding over libvfhip.so (include/vfhip.h) for tests, bench.py and Python users.

It mirrors the reference's renderer objects one-to-one: a class per element with
configure / process / cleanup, BOOL-style failures turned into VfHipError.  It contains NO compute:
every pixel is produced by the HIP kernels in libvfhip.so, and importing this module fails loudly when
the library is missing (there is no CPU fallback).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# $VFHIP_LIB: another build of the SAME library (tools/exp A/B variants) — never a fallback: the file must exist, and the choice is printed
LIB_PATH = os.environ.get("VFHIP_LIB") or os.path.join(_HERE, "libvfhip.so")

FORMATS = {"BGRA": 0, "RGBA": 1, "NV12": 2, "I420": 3, "UYVY": 4, "YUY2": 5}
MATRICES = {"bt601": 0, "bt709": 1, "bt2020": 2}
CHROMA_SITES = {"jpeg": 0, "none": 0, "center": 0, "mpeg2": 1}
METHODS = {"bilinear": 0, "nearest": 1, "bicubic": 2}
NUMERICS = {"gst-exact": 0, "metal": 1, "gst-exact-strict": 2}
FRAME_FLAG_TFF = 1


class VfHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"vfhip error {code}: {msg}")
        self.code = code


class VideoInfo(C.Structure):
    _fields_ = [("format", C.c_int32), ("width", C.c_int32), ("height", C.c_int32), ("color_matrix", C.c_int32),
                ("chroma_site", C.c_int32), ("reserved", C.c_int32 * 3)]


class Frame(C.Structure):
    _fields_ = [("info", VideoInfo), ("data", C.c_void_p * 4), ("stride", C.c_int32 * 4), ("flags", C.c_uint32),
                ("reserved", C.c_uint32)]


def _load():
    # PyTorch-ROCm wheels bundle their own libamdhip64.so.7; two HIP runtimes in one process do not share
    # devices ("No HIP GPUs are available").  Import torch first (when present) so that libvfhip binds to the
    # runtime already loaded; stand-alone users (the GStreamer plugin) get /opt/rocm's via RUNPATH.
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `make -C {_HERE}` (or __graft_entry__.build()); "
                          "vfhip has no CPU fallback")
    if os.environ.get("VFHIP_LIB"):
        import sys
        print(f"vfhip: using $VFHIP_LIB = {LIB_PATH} instead of the product library", file=sys.stderr)
    lib = C.CDLL(LIB_PATH)
    lib.vfhip_last_error_string.restype = C.c_char_p
    for n in ("vfhip_pinned_alloc", "vfhip_device_malloc", "vfhip_convertscale_new", "vfhip_deinterlace_new",
              "vfhip_videofilter_new", "vfhip_compositor_new", "vfhip_transform_new", "vfhip_overlay_new"):
        getattr(lib, n).restype = C.c_void_p
    lib.vfhip_convertscale_kernel_name.restype = C.c_char_p
    lib.vfhip_pinned_alloc.argtypes = [C.c_int, C.c_size_t]
    lib.vfhip_pinned_free.argtypes = [C.c_void_p]
    lib.vfhip_device_malloc.argtypes = [C.c_int, C.c_size_t]
    lib.vfhip_device_free.argtypes = [C.c_int, C.c_void_p]
    lib.vfhip_memcpy_h2d.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t]
    lib.vfhip_memcpy_d2h.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t]
    lib.vfhip_convertscale_configure.argtypes = [C.c_void_p, C.POINTER(VideoInfo), C.POINTER(VideoInfo), C.c_int, C.c_int,
                                                 C.c_uint32, C.c_int]
    lib.vfhip_convertscale_process.argtypes = [C.c_void_p, C.POINTER(Frame), C.POINTER(Frame)]
    lib.vfhip_convertscale_submit.argtypes = [C.c_void_p, C.POINTER(Frame), C.POINTER(Frame)]
    lib.vfhip_convertscale_wait.argtypes = [C.c_void_p]
    lib.vfhip_convertscale_in_flight.argtypes = [C.c_void_p]
    lib.vfhip_convertscale_process_device.argtypes = [C.c_void_p, C.POINTER(Frame), C.POINTER(Frame), C.c_void_p]
    lib.vfhip_convertscale_process_device_batch.argtypes = [C.c_void_p, C.POINTER(Frame), C.POINTER(Frame), C.c_size_t,
                                                            C.c_size_t, C.c_int, C.c_void_p]
    for n in ("vfhip_convertscale_cleanup", "vfhip_convertscale_free", "vfhip_convertscale_kernel_name", "vfhip_convertscale_numerics_in_effect"):
        getattr(lib, n).argtypes = [C.c_void_p]
    lib.vfhip_deinterlace_configure.argtypes = [C.c_void_p, C.POINTER(VideoInfo)]
    lib.vfhip_deinterlace_process.argtypes = [C.c_void_p, C.POINTER(Frame), C.POINTER(Frame), C.POINTER(DeinterlaceParams)]
    lib.vfhip_deinterlace_process_device.argtypes = [C.c_void_p, C.POINTER(Frame), C.POINTER(Frame), C.POINTER(DeinterlaceParams), C.c_void_p]
    lib.vfhip_videofilter_configure.argtypes = [C.c_void_p, C.POINTER(VideoInfo), C.POINTER(VideoInfo)]
    lib.vfhip_videofilter_process.argtypes = [C.c_void_p, C.POINTER(Frame), C.POINTER(Frame), C.POINTER(VideoFilterParams)]
    lib.vfhip_videofilter_process_device.argtypes = [C.c_void_p, C.POINTER(Frame), C.POINTER(Frame), C.POINTER(VideoFilterParams), C.c_void_p]
    lib.vfhip_videofilter_load_lut.argtypes = [C.c_void_p, C.c_char_p]
    lib.vfhip_videofilter_set_lut.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    lib.vfhip_compositor_configure.argtypes = [C.c_void_p, C.POINTER(VideoInfo)]
    lib.vfhip_compositor_composite.argtypes = [C.c_void_p, C.POINTER(PadInput), C.c_int, C.c_int, C.POINTER(Frame)]
    lib.vfhip_compositor_submit.argtypes = lib.vfhip_compositor_composite.argtypes
    lib.vfhip_compositor_wait.argtypes = [C.c_void_p]
    lib.vfhip_compositor_in_flight.argtypes = [C.c_void_p]
    lib.vfhip_compositor_composite_device.argtypes = [C.c_void_p, C.POINTER(PadInput), C.c_int, C.c_int, C.POINTER(Frame), C.c_void_p]
    lib.vfhip_transform_configure.argtypes = [C.c_void_p, C.POINTER(VideoInfo), C.POINTER(VideoInfo)]
    lib.vfhip_transform_process.argtypes = [C.c_void_p, C.POINTER(Frame), C.POINTER(Frame), C.POINTER(TransformParams)]
    lib.vfhip_transform_process_device.argtypes = [C.c_void_p, C.POINTER(Frame), C.POINTER(Frame), C.POINTER(TransformParams), C.c_void_p]
    batch = [C.c_void_p, C.POINTER(Frame), C.POINTER(Frame), C.c_size_t, C.c_size_t, C.c_int]
    lib.vfhip_deinterlace_process_device_batch.argtypes = batch + [C.POINTER(DeinterlaceParams), C.c_void_p]
    lib.vfhip_videofilter_process_device_batch.argtypes = batch + [C.POINTER(VideoFilterParams), C.c_void_p]
    lib.vfhip_transform_process_device_batch.argtypes = batch + [C.POINTER(TransformParams), C.c_void_p]
    lib.vfhip_compositor_composite_device_batch.argtypes = [C.c_void_p, C.POINTER(PadInput), C.POINTER(C.c_size_t), C.c_int, C.c_int,
                                                            C.POINTER(Frame), C.c_size_t, C.c_int, C.c_void_p]
    lib.vfhip_overlay_configure.argtypes = [C.c_void_p, C.POINTER(VideoInfo), C.POINTER(VideoInfo)]
    lib.vfhip_overlay_load_image.argtypes = [C.c_void_p, C.c_char_p]
    lib.vfhip_overlay_set_image.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    lib.vfhip_overlay_image_size.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.vfhip_overlay_process.argtypes = [C.c_void_p, C.POINTER(Frame), C.POINTER(Frame), C.POINTER(OverlayParams)]
    lib.vfhip_overlay_process_device_batch.argtypes = batch + [C.POINTER(OverlayParams), C.c_void_p]
    for n in ("vfhip_overlay_clear_image", "vfhip_overlay_cleanup", "vfhip_overlay_free"):
        getattr(lib, n).argtypes = [C.c_void_p]
    lib.vfhip_transform_cleanup.argtypes = [C.c_void_p]
    lib.vfhip_transform_free.argtypes = [C.c_void_p]
    for n in ("vfhip_deinterlace_reset", "vfhip_deinterlace_cleanup", "vfhip_deinterlace_free", "vfhip_videofilter_clear_lut",
              "vfhip_videofilter_lut_size", "vfhip_videofilter_cleanup", "vfhip_videofilter_free", "vfhip_compositor_cleanup",
              "vfhip_compositor_free"):
        getattr(lib, n).argtypes = [C.c_void_p]
    return lib


class DeinterlaceParams(C.Structure):
    _fields_ = [("method", C.c_int32), ("top_field_first", C.c_int32), ("motion_threshold", C.c_float), ("reserved", C.c_int32)]


class VideoFilterParams(C.Structure):
    _fields_ = [("brightness", C.c_float), ("contrast", C.c_float), ("saturation", C.c_float), ("hue", C.c_float),
                ("gamma", C.c_float), ("sharpness", C.c_float), ("sepia", C.c_float), ("noise", C.c_float),
                ("vignette", C.c_float), ("invert", C.c_int32), ("chroma_key_enabled", C.c_int32),
                ("chroma_key_r", C.c_float), ("chroma_key_g", C.c_float), ("chroma_key_b", C.c_float),
                ("chroma_key_tolerance", C.c_float), ("chroma_key_smoothness", C.c_float), ("frame_index", C.c_uint32)]


class TransformParams(C.Structure):
    _fields_ = [("method", C.c_int32), ("crop_top", C.c_int32), ("crop_bottom", C.c_int32), ("crop_left", C.c_int32),
                ("crop_right", C.c_int32), ("reserved", C.c_int32 * 3)]


class OverlayParams(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("width", C.c_float), ("height", C.c_float), ("alpha", C.c_float)]


class PadInput(C.Structure):
    _fields_ = [("frame", Frame), ("xpos", C.c_int32), ("ypos", C.c_int32), ("width", C.c_int32), ("height", C.c_int32),
                ("alpha", C.c_double), ("blend_mode", C.c_int32), ("reserved", C.c_int32)]


lib = _load()


def check(rc):
    if rc < 0:
        raise VfHipError(rc, lib.vfhip_last_error_string().decode(errors="replace"))
    return rc


def device_count():
    return check(lib.vfhip_device_count())


def device_name(device=0):
    buf = C.create_string_buffer(256)
    check(lib.vfhip_device_name(device, buf, 256))
    return buf.value.decode()


def make_info(fmt, w, h, colorimetry="bt601", chroma_site="jpeg"):
    return VideoInfo(FORMATS[fmt], w, h, MATRICES[colorimetry], CHROMA_SITES[chroma_site])


def r4(x):
    return (x + 3) // 4 * 4


def plane_layout(fmt, w, h):
    """GstVideoInfo default layout -> [(offset, stride, rows)], total size."""
    hp = (h + 1) // 2 * 2
    if fmt in ("BGRA", "RGBA"):
        return [(0, 4 * w, h)], 4 * w * h
    if fmt in ("UYVY", "YUY2"):
        s = r4(2 * w)
        return [(0, s, h)], s * h
    if fmt == "NV12":
        s = r4(w)
        return [(0, s, h), (s * hp, s, hp // 2)], s * hp + s * (hp // 2)
    if fmt == "I420":
        s, cs = r4(w), r4((w + 1) // 2)
        uo = s * hp
        vo = uo + cs * (hp // 2)
        return [(0, s, h), (uo, cs, hp // 2), (vo, cs, hp // 2)], vo + cs * (hp // 2)
    raise ValueError(fmt)


def frame_from_base(info, fmt, w, h, base_ptr, flags=0, layout=None):
    """Frame whose planes live at base_ptr + GstVideoInfo-default offsets (host or device pointer)."""
    f = Frame()
    f.info = info
    f.flags = flags
    pl, _ = layout or plane_layout(fmt, w, h)
    for i, (off, stride, _rows) in enumerate(pl):
        f.data[i] = base_ptr + off
        f.stride[i] = stride
    return f


class ConvertScale:
    """MetalConvertScaleRenderer equivalent (reference convertscale/metalconvertscalerenderer.h:35-50)."""

    def __init__(self, device=-1):
        self.h = lib.vfhip_convertscale_new(device)
        if not self.h:
            raise VfHipError(-6, lib.vfhip_last_error_string().decode(errors="replace"))
        self.cfg = None

    def configure(self, in_fmt, in_w, in_h, out_fmt, out_w, out_h, method="bilinear", add_borders=False,
                  border_color=0xFF000000, numerics="gst-exact", colorimetry="bt601", chroma_site="jpeg",
                  out_colorimetry=None, out_chroma_site=None):
        self.in_info = make_info(in_fmt, in_w, in_h, colorimetry, chroma_site)
        self.out_info = make_info(out_fmt, out_w, out_h, out_colorimetry or colorimetry, out_chroma_site or chroma_site)
        check(lib.vfhip_convertscale_configure(self.h, C.byref(self.in_info), C.byref(self.out_info), METHODS[method],
                                               int(add_borders), border_color, NUMERICS[numerics]))
        self.cfg = (in_fmt, in_w, in_h, out_fmt, out_w, out_h)
        return self

    @property
    def kernel_name(self):
        return lib.vfhip_convertscale_kernel_name(self.h).decode()

    @property
    def numerics_in_effect(self):
        """'gst-exact' or 'metal': what the configured cell really computes (differs from the request on unpinned cells)"""
        rc = lib.vfhip_convertscale_numerics_in_effect(self.h)
        check(min(rc, 0))
        return {0: "gst-exact", 1: "metal"}[rc]

    def process(self, raw_in):
        """raw_in: uint8 array in GstVideoInfo default layout (host). Returns the raw output frame (host)."""
        in_fmt, in_w, in_h, out_fmt, out_w, out_h = self.cfg
        raw_in = np.ascontiguousarray(raw_in, dtype=np.uint8)
        _, in_size = plane_layout(in_fmt, in_w, in_h)
        assert raw_in.size >= in_size, (raw_in.size, in_size)
        _, out_size = plane_layout(out_fmt, out_w, out_h)
        out = np.zeros(out_size, np.uint8)
        fi = frame_from_base(self.in_info, in_fmt, in_w, in_h, raw_in.ctypes.data)
        fo = frame_from_base(self.out_info, out_fmt, out_w, out_h, out.ctypes.data)
        check(lib.vfhip_convertscale_process(self.h, C.byref(fi), C.byref(fo)))
        return out

    def process_device(self, in_ptr, out_ptr, stream=None, n_frames=1, in_pitch=0, out_pitch=0, in_layout=None, out_layout=None):
        in_fmt, in_w, in_h, out_fmt, out_w, out_h = self.cfg
        fi = frame_from_base(self.in_info, in_fmt, in_w, in_h, in_ptr, layout=in_layout)
        fo = frame_from_base(self.out_info, out_fmt, out_w, out_h, out_ptr, layout=out_layout)
        check(lib.vfhip_convertscale_process_device_batch(self.h, C.byref(fi), C.byref(fo), in_pitch, out_pitch, n_frames,
                                                          stream))

    def cleanup(self):
        lib.vfhip_convertscale_cleanup(self.h)

    def close(self):
        if self.h:
            lib.vfhip_convertscale_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


DEINTERLACE_METHODS = {"bob": 0, "weave": 1, "linear": 2, "greedyh": 3}
BLEND_MODES = {"source": 0, "over": 1, "add": 2}
BACKGROUNDS = {"checker": 0, "black": 1, "white": 2, "transparent": 3}


def filter_params(brightness=0.0, contrast=1.0, saturation=1.0, hue=0.0, gamma=1.0, sharpness=0.0, sepia=0.0, noise=0.0,
                  vignette=0.0, invert=False, chroma_key=None, tolerance=0.2, smoothness=0.1, frame_index=0):
    """Defaults = the element's property defaults (reference videofilter/gstvfmetalvideofilter.m:435-533).
    hue is in radians here (the element multiplies its [-1,1] property by pi, :189)."""
    p = VideoFilterParams(brightness, contrast, saturation, hue, gamma, sharpness, sepia, noise, vignette, int(invert),
                          int(chroma_key is not None), 0.0, 0.0, 0.0, tolerance, smoothness, frame_index)
    if chroma_key is not None:
        p.chroma_key_r, p.chroma_key_g, p.chroma_key_b = chroma_key
    return p


class _Element:
    _free = None

    def close(self):
        if getattr(self, "h", None):
            getattr(lib, self._free)(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Deinterlace(_Element):
    """MetalDeinterlaceRenderer equivalent (reference deinterlace/metaldeinterlacerenderer.h:42-54)."""
    _free = "vfhip_deinterlace_free"

    def __init__(self, device=-1):
        self.h = lib.vfhip_deinterlace_new(device)
        if not self.h:
            raise VfHipError(-6, lib.vfhip_last_error_string().decode(errors="replace"))

    def configure(self, fmt, w, h, colorimetry="bt601"):
        self.fmt, self.w, self.hh = fmt, w, h
        self.info = make_info(fmt, w, h, colorimetry)
        check(lib.vfhip_deinterlace_configure(self.h, C.byref(self.info)))
        return self

    def process(self, raw_in, method="bob", tff=True, threshold=0.1):
        raw_in = np.ascontiguousarray(raw_in, dtype=np.uint8)
        _, size = plane_layout(self.fmt, self.w, self.hh)
        out = np.zeros(size, np.uint8)
        fi = frame_from_base(self.info, self.fmt, self.w, self.hh, raw_in.ctypes.data)
        fo = frame_from_base(self.info, self.fmt, self.w, self.hh, out.ctypes.data)
        prm = DeinterlaceParams(DEINTERLACE_METHODS[method], int(tff), threshold, 0)
        check(lib.vfhip_deinterlace_process(self.h, C.byref(fi), C.byref(fo), C.byref(prm)))
        return out

    def process_device(self, in_ptr, out_ptr, method="bob", tff=True, threshold=0.1, stream=None, n_frames=1, in_pitch=0, out_pitch=0,
                       in_layout=None, out_layout=None):
        """n_frames > 1: consecutive frames of one stream, frame k at ptr + k * pitch (history of k = frame k-1)"""
        fi = frame_from_base(self.info, self.fmt, self.w, self.hh, in_ptr, layout=in_layout)
        fo = frame_from_base(self.info, self.fmt, self.w, self.hh, out_ptr, layout=out_layout)
        prm = DeinterlaceParams(DEINTERLACE_METHODS[method], int(tff), threshold, 0)
        check(lib.vfhip_deinterlace_process_device_batch(self.h, C.byref(fi), C.byref(fo), in_pitch, out_pitch, n_frames, C.byref(prm), stream))

    def reset(self):
        check(lib.vfhip_deinterlace_reset(self.h))


class VideoFilter(_Element):
    """MetalVideoFilterRenderer equivalent (reference videofilter/metalvideofilterrenderer.h:48-70)."""
    _free = "vfhip_videofilter_free"

    def __init__(self, device=-1):
        self.h = lib.vfhip_videofilter_new(device)
        if not self.h:
            raise VfHipError(-6, lib.vfhip_last_error_string().decode(errors="replace"))

    def configure(self, in_fmt, w, h, out_fmt=None, colorimetry="bt601"):
        self.in_fmt, self.out_fmt, self.w, self.hh = in_fmt, out_fmt or in_fmt, w, h
        self.in_info = make_info(in_fmt, w, h, colorimetry)
        self.out_info = make_info(self.out_fmt, w, h, colorimetry)
        check(lib.vfhip_videofilter_configure(self.h, C.byref(self.in_info), C.byref(self.out_info)))
        return self

    def process(self, raw_in, params):
        raw_in = np.ascontiguousarray(raw_in, dtype=np.uint8)
        _, size = plane_layout(self.out_fmt, self.w, self.hh)
        out = np.zeros(size, np.uint8)
        fi = frame_from_base(self.in_info, self.in_fmt, self.w, self.hh, raw_in.ctypes.data)
        fo = frame_from_base(self.out_info, self.out_fmt, self.w, self.hh, out.ctypes.data)
        check(lib.vfhip_videofilter_process(self.h, C.byref(fi), C.byref(fo), C.byref(params)))
        return out

    def process_device(self, in_ptr, out_ptr, params, stream=None, n_frames=1, in_pitch=0, out_pitch=0, in_layout=None, out_layout=None):
        """n_frames > 1: frame k at ptr + k * pitch, filtered with frame_index + k"""
        fi = frame_from_base(self.in_info, self.in_fmt, self.w, self.hh, in_ptr, layout=in_layout)
        fo = frame_from_base(self.out_info, self.out_fmt, self.w, self.hh, out_ptr, layout=out_layout)
        check(lib.vfhip_videofilter_process_device_batch(self.h, C.byref(fi), C.byref(fo), in_pitch, out_pitch, n_frames, C.byref(params), stream))

    def load_lut(self, path):
        check(lib.vfhip_videofilter_load_lut(self.h, path.encode()))

    def set_lut(self, rgba):
        rgba = np.ascontiguousarray(rgba, dtype=np.float32)
        size = int(round((rgba.size // 4) ** (1 / 3)))
        assert size ** 3 * 4 == rgba.size
        check(lib.vfhip_videofilter_set_lut(self.h, rgba.ctypes.data, size))

    def clear_lut(self):
        lib.vfhip_videofilter_clear_lut(self.h)

    @property
    def lut_size(self):
        return lib.vfhip_videofilter_lut_size(self.h)


class Compositor(_Element):
    """MetalCompositorRenderer equivalent (reference compositor/metalcomprenderer.h:51-63)."""
    _free = "vfhip_compositor_free"

    def __init__(self, device=-1):
        self.h = lib.vfhip_compositor_new(device)
        if not self.h:
            raise VfHipError(-6, lib.vfhip_last_error_string().decode(errors="replace"))

    def configure(self, fmt, w, h, colorimetry="bt601"):
        self.fmt, self.w, self.hh = fmt, w, h
        self.info = make_info(fmt, w, h, colorimetry)
        check(lib.vfhip_compositor_configure(self.h, C.byref(self.info)))
        return self

    @staticmethod
    def pad(fmt, w, h, base_ptr, xpos, ypos, width, height, alpha=1.0, blend="over", colorimetry="bt601", layout=None):
        p = PadInput()
        p.frame = frame_from_base(make_info(fmt, w, h, colorimetry), fmt, w, h, base_ptr, layout=layout)
        p.xpos, p.ypos, p.width, p.height, p.alpha, p.blend_mode = xpos, ypos, width, height, alpha, BLEND_MODES[blend]
        return p

    def composite(self, pads, background="checker"):
        """pads: list of (fmt, w, h, raw ndarray, xpos, ypos, width, height, alpha, blend[, colorimetry])."""
        keep = [np.ascontiguousarray(p[3], dtype=np.uint8) for p in pads]
        arr = (PadInput * max(len(pads), 1))()
        for i, p in enumerate(pads):
            arr[i] = self.pad(p[0], p[1], p[2], keep[i].ctypes.data, *p[4:])
        _, size = plane_layout(self.fmt, self.w, self.hh)
        out = np.zeros(size, np.uint8)
        fo = frame_from_base(self.info, self.fmt, self.w, self.hh, out.ctypes.data)
        check(lib.vfhip_compositor_composite(self.h, arr, len(pads), BACKGROUNDS[background], C.byref(fo)))
        return out

    def composite_device(self, pad_structs, out_ptr, background="checker", stream=None, n_frames=1, pad_pitches=None, out_pitch=0, out_layout=None):
        """n_frames > 1: pad i's frame k at its base + k * pad_pitches[i], output frame k at out_ptr + k * out_pitch"""
        arr = (PadInput * max(len(pad_structs), 1))(*pad_structs)
        fo = frame_from_base(self.info, self.fmt, self.w, self.hh, out_ptr, layout=out_layout)
        if n_frames == 1 and pad_pitches is None:
            check(lib.vfhip_compositor_composite_device(self.h, arr, len(pad_structs), BACKGROUNDS[background], C.byref(fo), stream))
            return
        pit = (C.c_size_t * max(len(pad_structs), 1))(*(pad_pitches or [0] * len(pad_structs)))
        check(lib.vfhip_compositor_composite_device_batch(self.h, arr, pit, len(pad_structs), BACKGROUNDS[background], C.byref(fo), out_pitch,
                                                          n_frames, stream))


TRANSFORM_METHODS = {"none": 0, "clockwise": 1, "rotate-180": 2, "counterclockwise": 3, "horizontal-flip": 4, "vertical-flip": 5,
                     "upper-left-diagonal": 6, "upper-right-diagonal": 7}


class Transform(_Element):
    """MetalTransformRenderer equivalent (reference transform/metaltransformrenderer.h)."""
    _free = "vfhip_transform_free"

    def __init__(self, device=-1):
        self.h = lib.vfhip_transform_new(device)
        if not self.h:
            raise VfHipError(-6, lib.vfhip_last_error_string().decode(errors="replace"))

    def configure(self, in_fmt, w, h, out_fmt=None, colorimetry="bt601"):
        self.in_fmt, self.out_fmt, self.w, self.hh = in_fmt, out_fmt or in_fmt, w, h
        self.in_info = make_info(in_fmt, w, h, colorimetry)
        self.out_info = make_info(self.out_fmt, w, h, colorimetry)
        check(lib.vfhip_transform_configure(self.h, C.byref(self.in_info), C.byref(self.out_info)))
        return self

    def process(self, raw_in, method="none", crop=(0, 0, 0, 0)):
        """crop = (top, bottom, left, right)"""
        raw_in = np.ascontiguousarray(raw_in, dtype=np.uint8)
        out = np.zeros(plane_layout(self.out_fmt, self.w, self.hh)[1], np.uint8)
        fi = frame_from_base(self.in_info, self.in_fmt, self.w, self.hh, raw_in.ctypes.data)
        fo = frame_from_base(self.out_info, self.out_fmt, self.w, self.hh, out.ctypes.data)
        prm = TransformParams(TRANSFORM_METHODS[method], *crop)
        check(lib.vfhip_transform_process(self.h, C.byref(fi), C.byref(fo), C.byref(prm)))
        return out

    def process_device(self, in_ptr, out_ptr, method="none", crop=(0, 0, 0, 0), stream=None, n_frames=1, in_pitch=0, out_pitch=0,
                       in_layout=None, out_layout=None):
        fi = frame_from_base(self.in_info, self.in_fmt, self.w, self.hh, in_ptr, layout=in_layout)
        fo = frame_from_base(self.out_info, self.out_fmt, self.w, self.hh, out_ptr, layout=out_layout)
        prm = TransformParams(TRANSFORM_METHODS[method], *crop)
        check(lib.vfhip_transform_process_device_batch(self.h, C.byref(fi), C.byref(fo), in_pitch, out_pitch, n_frames, C.byref(prm), stream))


class Overlay(_Element):
    """MetalOverlayRenderer equivalent (reference overlay/metaloverlayrenderer.h)."""
    _free = "vfhip_overlay_free"

    def __init__(self, device=-1):
        self.h = lib.vfhip_overlay_new(device)
        if not self.h:
            raise VfHipError(-6, lib.vfhip_last_error_string().decode(errors="replace"))

    def configure(self, in_fmt, w, h, out_fmt=None, colorimetry="bt601"):
        self.in_fmt, self.out_fmt, self.w, self.hh = in_fmt, out_fmt or in_fmt, w, h
        self.in_info = make_info(in_fmt, w, h, colorimetry)
        self.out_info = make_info(self.out_fmt, w, h, colorimetry)
        check(lib.vfhip_overlay_configure(self.h, C.byref(self.in_info), C.byref(self.out_info)))
        return self

    def load_image(self, path):
        check(lib.vfhip_overlay_load_image(self.h, (path or "").encode()))

    def set_image(self, rgba):
        """rgba: (h, w, 4) uint8, the bytes the shader sees (the reference's texture is premultiplied)"""
        rgba = np.ascontiguousarray(rgba, np.uint8)
        check(lib.vfhip_overlay_set_image(self.h, rgba.ctypes.data, rgba.shape[1], rgba.shape[0]))

    def clear_image(self):
        lib.vfhip_overlay_clear_image(self.h)

    @property
    def image_size(self):
        w, h = C.c_int(), C.c_int()
        return (w.value, h.value) if lib.vfhip_overlay_image_size(self.h, C.byref(w), C.byref(h)) else None

    def process(self, raw_in, x=0.0, y=0.0, width=0.0, height=0.0, alpha=1.0):
        raw_in = np.ascontiguousarray(raw_in, dtype=np.uint8)
        out = np.zeros(plane_layout(self.out_fmt, self.w, self.hh)[1], np.uint8)
        fi = frame_from_base(self.in_info, self.in_fmt, self.w, self.hh, raw_in.ctypes.data)
        fo = frame_from_base(self.out_info, self.out_fmt, self.w, self.hh, out.ctypes.data)
        prm = OverlayParams(x, y, width, height, alpha)
        check(lib.vfhip_overlay_process(self.h, C.byref(fi), C.byref(fo), C.byref(prm)))
        return out

    def process_device(self, in_ptr, out_ptr, x=0.0, y=0.0, width=0.0, height=0.0, alpha=1.0, stream=None, n_frames=1, in_pitch=0, out_pitch=0,
                       in_layout=None, out_layout=None):
        fi = frame_from_base(self.in_info, self.in_fmt, self.w, self.hh, in_ptr, layout=in_layout)
        fo = frame_from_base(self.out_info, self.out_fmt, self.w, self.hh, out_ptr, layout=out_layout)
        prm = OverlayParams(x, y, width, height, alpha)
        check(lib.vfhip_overlay_process_device_batch(self.h, C.byref(fi), C.byref(fo), in_pitch, out_pitch, n_frames, C.byref(prm), stream))
