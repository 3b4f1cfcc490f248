"""End-to-end drop-in tests on the GPU box: real gst-launch-1.0 pipelines through the vfhip elements.

  * the reference's smoke matrix (tests/test-convertscale.sh, test-videofilter.sh, test-deinterlace.sh,
    test-multi-element.sh) re-targeted to the vfhip* names: exit status only, like the reference;
  * what the reference never had — pixel parity: the same videotestsrc frames through GStreamer's CPU
    `videoconvert ! videoscale` and through `vfhipconvertscale`, compared byte for byte (BASELINE config[0]).
Skipped when the box has no /opt/conda GStreamer."""
import os

import numpy as np
import pytest

import gst_env

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not gst_env.available(), reason="GStreamer 1.14 (/opt/conda) or libgstvfhip.so not present")]

SRC = "videotestsrc num-buffers=3"


def ok(pipeline):
    r = gst_env.launch(pipeline)
    assert r.returncode == 0, f"{pipeline}\n{r.stdout}\n{r.stderr}"


def caps(fmt, w, h, extra=""):
    return f"video/x-raw,format={fmt},width={w},height={h}{extra}"


@pytest.mark.parametrize("fmt", ["BGRA", "RGBA", "NV12", "I420"])
def test_convertscale_passthrough(fmt):
    ok(f"{SRC} ! {caps(fmt, 320, 240)} ! vfhipconvertscale ! {caps(fmt, 320, 240)} ! fakesink")


@pytest.mark.parametrize("ifmt", ["BGRA", "RGBA", "NV12", "I420", "UYVY", "YUY2"])
@pytest.mark.parametrize("ofmt", ["BGRA", "NV12", "I420", "UYVY"])
def test_convertscale_conversions(ifmt, ofmt):
    ok(f"{SRC} ! {caps(ifmt, 320, 240)} ! vfhipconvertscale ! {caps(ofmt, 320, 240)} ! fakesink")


@pytest.mark.parametrize("p", [
    f"{caps('BGRA', 320, 240)} ! vfhipconvertscale ! {caps('BGRA', 640, 480)}",
    f"{caps('BGRA', 640, 480)} ! vfhipconvertscale method=nearest ! {caps('BGRA', 160, 120)}",
    f"{caps('NV12', 1920, 1080)} ! vfhipconvertscale ! {caps('BGRA', 640, 480)}",
    f"{caps('I420', 1280, 720)} ! vfhipconvertscale numerics=metal ! {caps('NV12', 640, 360)}",
    f"{caps('BGRA', 640, 360)} ! vfhipconvertscale add-borders=true border-color=0xFF102030 ! {caps('BGRA', 320, 320)}",
    f"{caps('BGRA', 17, 13)} ! vfhipconvertscale ! {caps('BGRA', 33, 7)}",
    f"{caps('NV12', 64, 36)} ! vfhipconvertscale ! video/x-raw,format=BGRA,width=32",        # height from the DAR
])
def test_convertscale_scaling(p):
    ok(f"{SRC} ! {p} ! fakesink")


@pytest.mark.parametrize("props", ["brightness=0.3", "contrast=1.5", "saturation=0.2", "hue=0.5", "gamma=2.2", "sharpness=0.8", "sharpness=-0.8",
                                   "sepia=1.0", "invert=true", "noise=0.5", "vignette=0.8",
                                   "chroma-key-enabled=true chroma-key-color=0xFF00FF00 chroma-key-tolerance=0.3 chroma-key-smoothness=0.1",
                                   "brightness=0.1 contrast=1.2 saturation=0.8 hue=0.3 gamma=1.5 sharpness=0.5 sepia=0.2 noise=0.1 vignette=0.3", ""])
def test_videofilter_properties(props):
    ok(f"{SRC} ! {caps('BGRA', 320, 240)} ! vfhipvideofilter {props} ! fakesink")


@pytest.mark.parametrize("fmt,size", [("NV12", (320, 240)), ("I420", (320, 240)), ("RGBA", (321, 241)), ("BGRA", (1920, 1080))])
def test_videofilter_formats(fmt, size):
    ok(f"{SRC} ! {caps(fmt, *size)} ! vfhipvideofilter brightness=0.2 sharpness=0.4 ! fakesink")


def test_videofilter_lut_file(tmp_path):
    n = 4
    rows = [f"LUT_3D_SIZE {n}"] + ["%f %f %f" % (r / (n - 1), g / (n - 1), b / (n - 1)) for b in range(n) for g in range(n) for r in range(n)]
    p = tmp_path / "id.cube"
    p.write_text("\n".join(rows) + "\n")
    ok(f"{SRC} ! {caps('BGRA', 320, 240)} ! vfhipvideofilter lut-file={p} ! fakesink")
    ok(f"{SRC} ! {caps('BGRA', 320, 240)} ! vfhipvideofilter lut-file={tmp_path / 'missing.cube'} saturation=0.5 ! fakesink")   # warning, not an error


@pytest.mark.parametrize("method", ["bob", "weave", "linear", "greedyh"])
@pytest.mark.parametrize("fmt", ["BGRA", "NV12", "I420"])
def test_deinterlace_methods(method, fmt):
    ok(f"videotestsrc num-buffers=4 pattern=ball ! {caps(fmt, 320, 240)} ! vfhipdeinterlace method={method} field-layout=top-field-first ! fakesink")


def test_deinterlace_field_layout_and_1080p():
    ok(f"{SRC} ! {caps('NV12', 1920, 1080)} ! vfhipdeinterlace method=greedyh field-layout=bottom-field-first motion-threshold=0.05 ! fakesink")
    ok(f"{SRC} ! {caps('BGRA', 321, 241)} ! vfhipdeinterlace method=weave ! fakesink")


@pytest.mark.parametrize("method", ["clockwise", "rotate-180", "counterclockwise", "horizontal-flip", "vertical-flip", "upper-left-diagonal", "upper-right-diagonal", "none"])
def test_transform_methods(method):
    ok(f"{SRC} ! {caps('BGRA', 320, 240)} ! vfhiptransform method={method} ! fakesink")
    ok(f"{SRC} ! {caps('NV12', 320, 240)} ! vfhiptransform method={method} crop-top=10 crop-left=20 ! fakesink")


def test_multi_element_chains():
    """several renderers (each with its own stream trio) in one process (reference tests/test-multi-element.sh:2-4)"""
    ok(f"{SRC} ! {caps('NV12', 640, 480)} ! vfhipdeinterlace method=greedyh ! vfhipconvertscale ! {caps('BGRA', 320, 240)} ! vfhipvideofilter brightness=0.1 sepia=0.5 ! vfhiptransform method=horizontal-flip ! fakesink")
    ok(f"{SRC} ! {caps('BGRA', 640, 480)} ! tee name=t t. ! queue ! vfhipconvertscale ! {caps('NV12', 320, 240)} ! fakesink "
       f"t. ! queue ! vfhipvideofilter invert=true ! vfhipconvertscale ! {caps('I420', 160, 120)} ! fakesink")


@pytest.mark.parametrize("iw,ih,ow,oh,fmt", [(1920, 1080, 640, 480, "NV12"), (3840, 2160, 1920, 1080, "NV12"), (1280, 720, 1920, 1080, "I420"), (720, 576, 360, 288, "NV12"),
                                             # round 2's new kernels inside real pipelines: conversion at the same size (k_cs_yuv_same: NV12, I420, UYVY, YUY2),
                                             # up-scales and NV12 / UYVY down-scales through k_cs_bilinear_tile
                                             (1920, 1080, 1920, 1080, "NV12"), (1280, 720, 1280, 720, "I420"), (1280, 720, 1280, 720, "UYVY"), (640, 480, 640, 480, "YUY2"),
                                             (1280, 720, 1920, 1080, "NV12"), (1920, 1080, 1280, 720, "NV12"), (1920, 1080, 1280, 720, "UYVY"), (640, 360, 1280, 720, "YUY2")])
def test_pixel_parity_with_cpu_videoconvert_videoscale(tmp_path, iw, ih, ow, oh, fmt):
    """BASELINE configs[0] and [1] — and the shapes of round 2's kernels — as real pipelines: vfhipconvertscale is byte-identical to videoconvert ! videoscale"""
    a, b = tmp_path / "cpu.raw", tmp_path / "hip.raw"
    r = gst_env.launch(f"videotestsrc num-buffers=2 ! {caps(fmt, iw, ih)} ! tee name=t "
                       f"t. ! queue ! videoconvert ! videoscale ! {caps('BGRA', ow, oh)} ! filesink location={a} "
                       f"t. ! queue ! vfhipconvertscale ! {caps('BGRA', ow, oh)} ! filesink location={b}", timeout=300)
    assert r.returncode == 0, r.stderr
    x, y = np.fromfile(a, np.uint8), np.fromfile(b, np.uint8)
    assert x.size == y.size == 2 * ow * oh * 4
    assert np.array_equal(x, y), f"max diff {np.abs(x.astype(int) - y.astype(int)).max()}, {(x != y).sum()} bytes differ"


# ---- compositor: reference tests/test-compositor.sh:90-187 re-targeted, plus pixel parity with the oracle -------------
VT = "videotestsrc num-buffers=5"


@pytest.mark.parametrize("fmt,size", [("BGRA", (320, 240)), ("RGBA", (320, 240)), ("NV12", (320, 240)), ("I420", (320, 240)), ("BGRA", (1920, 1080)), ("BGRA", (160, 120))])
def test_compositor_single_input(fmt, size):
    ok(f"{VT} ! {caps(fmt, *size)} ! vfhipcompositor ! fakesink")


@pytest.mark.parametrize("bg", ["checker", "black", "white", "transparent"])
def test_compositor_backgrounds(bg):
    ok(f"{VT} ! {caps('BGRA', 320, 240)} ! vfhipcompositor background={bg} ! fakesink")


@pytest.mark.parametrize("p", [
    f"vfhipcompositor name=comp sink_0::xpos=0 sink_0::ypos=0 sink_1::xpos=160 sink_1::ypos=120 sink_1::alpha=0.7 ! fakesink "
    f"{VT} ! {caps('BGRA', 320, 240)} ! comp. {VT} pattern=snow ! {caps('BGRA', 320, 240)} ! comp.",
    f"vfhipcompositor name=comp sink_0::operator=source sink_1::operator=over sink_1::xpos=50 sink_1::ypos=50 sink_1::alpha=0.8 "
    f"sink_2::operator=add sink_2::xpos=100 sink_2::ypos=100 sink_2::alpha=0.5 ! fakesink "
    f"{VT} ! {caps('BGRA', 320, 240)} ! comp. {VT} pattern=snow ! {caps('BGRA', 160, 120)} ! comp. {VT} pattern=smpte ! {caps('BGRA', 160, 120)} ! comp.",
    f"vfhipcompositor name=comp sink_0::zorder=1 sink_1::zorder=0 ! fakesink "
    f"{VT} ! {caps('BGRA', 320, 240)} ! comp. {VT} pattern=snow ! {caps('BGRA', 320, 240)} ! comp.",
    f"vfhipcompositor name=comp sink_0::sizing-policy=keep-aspect-ratio sink_0::width=200 sink_0::height=200 ! fakesink {VT} ! {caps('BGRA', 320, 240)} ! comp.",
    f"vfhipcompositor name=comp sink_1::xpos=160 sink_1::ypos=120 ! fakesink {VT} ! {caps('BGRA', 320, 240)} ! comp. {VT} pattern=snow ! {caps('NV12', 160, 120)} ! comp.",
    f"{VT} ! vfhipcompositor ! {caps('NV12', 320, 240)} ! fakesink",
    f"{VT} ! vfhipcompositor ! {caps('I420', 320, 240)} ! fakesink",
])
def test_compositor_multi_input(p):
    ok(p)


def test_compositor_pixels_match_oracle(tmp_path):
    """two positioned BGRA + NV12 inputs, over / add, black background: the frames the element writes are byte-identical
    to oracle/metalref.c's restatement run on the very frames the sources produced"""
    import oracle_lib
    mr = oracle_lib.load_metalref()
    a, b, o = tmp_path / "a.raw", tmp_path / "b.raw", tmp_path / "o.raw"
    r = gst_env.launch(f"vfhipcompositor name=comp background=black sink_1::xpos=100 sink_1::ypos=60 sink_1::alpha=0.6 sink_1::operator=add "
                       f"! {caps('BGRA', 320, 240)} ! filesink location={o} "
                       f"videotestsrc num-buffers=2 ! {caps('BGRA', 320, 240)} ! tee name=ta ta. ! queue ! comp. ta. ! queue ! filesink location={a} "
                       f"videotestsrc num-buffers=2 pattern=ball ! {caps('NV12', 160, 120)} ! tee name=tb tb. ! queue ! comp. tb. ! queue ! filesink location={b}")
    assert r.returncode == 0, r.stderr
    fa, fb, fo = np.fromfile(a, np.uint8), np.fromfile(b, np.uint8), np.fromfile(o, np.uint8)
    na, nb, no = 320 * 240 * 4, 160 * 120 * 3 // 2, 320 * 240 * 4
    assert fa.size == 2 * na and fb.size == 2 * nb and fo.size == 2 * no
    for k in range(2):
        want = mr.compositor("BGRA", 320, 240, [("BGRA", 320, 240, fa[k * na:(k + 1) * na], 0, 0, 320, 240, 1.0, 1),
                                                 ("NV12", 160, 120, fb[k * nb:(k + 1) * nb], 100, 60, 160, 120, 0.6, 2)], 1)
        got = fo[k * no:(k + 1) * no]
        assert np.array_equal(want, got), f"frame {k}: {(want != got).sum()} bytes differ, max {np.abs(want.astype(int) - got.astype(int)).max()}"


# ---- memory:HIPMemory: chained vfhip elements keep frames in HBM ------------------------------
def hipcaps(fmt, w, h):
    return f"'video/x-raw(memory:HIPMemory),format={fmt},width={w},height={h}'"


def test_hip_memory_negotiated_between_vfhip_elements():
    r = gst_env.launch(f"{SRC} ! {caps('NV12', 640, 480)} ! vfhipdeinterlace method=greedyh name=d ! vfhipconvertscale name=c ! {caps('BGRA', 320, 240)} ! fakesink", verbose=True)
    assert r.returncode == 0, r.stderr
    link = [l for l in r.stdout.splitlines() if "c.GstPad:sink: caps" in l]
    assert link and "video/x-raw(memory:HIPMemory)" in link[0], r.stdout[-2000:]
    src = [l for l in r.stdout.splitlines() if "c.GstPad:src: caps" in l]
    assert src and "memory:HIPMemory" not in src[0]              # the capsfilter without features means system memory


def test_hip_memory_chain_matches_system_memory_chain(tmp_path):
    """the same 4-element chain with device buffers between the elements and with system-memory caps forced between
    them writes identical bytes"""
    a, b = tmp_path / "hip.raw", tmp_path / "sys.raw"
    chain = ["vfhipdeinterlace method=greedyh field-layout=top-field-first", "vfhipconvertscale numerics=metal", "vfhipvideofilter brightness=0.1 sharpness=0.4",
             "vfhiptransform method=horizontal-flip"]
    mid = [caps('NV12', 640, 480), caps('RGBA', 320, 240), caps('RGBA', 320, 240)]
    hip = " ! ".join(chain[:1] + [f"{chain[1]} ! {hipcaps('RGBA', 320, 240)}"] + chain[2:])
    sysm = " ! ".join(x for pair in zip(chain, mid + [""]) for x in pair if x)
    src = f"videotestsrc num-buffers=4 pattern=ball ! {caps('NV12', 640, 480)}"
    for path, body in ((a, hip), (b, sysm)):
        r = gst_env.launch(f"{src} ! {body} ! {caps('RGBA', 320, 240)} ! filesink location={path}")
        assert r.returncode == 0, f"{body}\n{r.stderr}"
    x, y = np.fromfile(a, np.uint8), np.fromfile(b, np.uint8)
    assert x.size == y.size == 4 * 320 * 240 * 4
    assert np.array_equal(x, y)


def test_hip_memory_is_cpu_mappable(tmp_path):
    """a memory:HIPMemory buffer reaching an element that maps it for the CPU (filesink) reads back the frame"""
    a, b = tmp_path / "hip.raw", tmp_path / "sys.raw"
    r = gst_env.launch(f"videotestsrc num-buffers=2 ! {caps('NV12', 320, 240)} ! tee name=t "
                       f"t. ! queue ! vfhipconvertscale ! {hipcaps('BGRA', 160, 120)} ! filesink location={a} "
                       f"t. ! queue ! vfhipconvertscale ! {caps('BGRA', 160, 120)} ! filesink location={b}")
    assert r.returncode == 0, r.stderr
    x, y = np.fromfile(a, np.uint8), np.fromfile(b, np.uint8)
    assert x.size >= 2 * 160 * 120 * 4 and np.array_equal(x.reshape(2, -1)[:, :160 * 120 * 4], y.reshape(2, -1))


def test_hip_memory_to_fakesink_and_passthrough():
    ok(f"{SRC} ! {caps('NV12', 1920, 1080)} ! vfhipconvertscale ! {hipcaps('BGRA', 640, 360)} ! vfhipvideofilter ! vfhiptransform ! fakesink")
    ok(f"{SRC} ! {caps('I420', 320, 240)} ! vfhipdeinterlace ! vfhipdeinterlace method=weave ! vfhipconvertscale ! {caps('UYVY', 160, 120)} ! fakesink")


def test_compositor_takes_hip_memory_pads(tmp_path):
    """device buffers from upstream vfhip elements composite to the same bytes as system-memory ones"""
    a, b = tmp_path / "hip.raw", tmp_path / "sys.raw"
    for path, mid in ((a, hipcaps('BGRA', 160, 120)), (b, caps('BGRA', 160, 120))):
        r = gst_env.launch(f"vfhipcompositor name=comp background=black sink_1::xpos=80 sink_1::ypos=60 sink_1::alpha=0.5 ! {caps('BGRA', 320, 240)} ! filesink location={path} "
                           f"videotestsrc num-buffers=3 ! {caps('BGRA', 320, 240)} ! comp. "
                           f"videotestsrc num-buffers=3 pattern=ball ! {caps('NV12', 320, 240)} ! vfhipconvertscale numerics=metal ! {mid} ! comp.")
        assert r.returncode == 0, r.stderr
    x, y = np.fromfile(a, np.uint8), np.fromfile(b, np.uint8)
    assert x.size == y.size == 3 * 320 * 240 * 4 and np.array_equal(x, y)


# ---- async-depth=1: one frame in flight across buffers ----------------------------------------
@pytest.mark.parametrize("n", [1, 2, 5])
def test_async_depth_same_frames_in_order(tmp_path, n):
    """every frame comes out, in order, byte-identical to the synchronous element — including the ones still in flight at EOS"""
    a, b = tmp_path / "async.raw", tmp_path / "sync.raw"
    for path, depth in ((a, 1), (b, 0)):
        r = gst_env.launch(f"videotestsrc num-buffers={n} pattern=ball ! {caps('NV12', 640, 480)} ! vfhipconvertscale async-depth={depth} ! "
                           f"{caps('BGRA', 320, 240)} ! filesink location={path}")
        assert r.returncode == 0, r.stderr
    x, y = np.fromfile(a, np.uint8), np.fromfile(b, np.uint8)
    assert x.size == y.size == n * 320 * 240 * 4 and np.array_equal(x, y)


def test_async_depth_in_chains():
    ok(f"{SRC} ! {caps('NV12', 1920, 1080)} ! vfhipdeinterlace ! vfhipconvertscale async-depth=1 ! {caps('BGRA', 640, 360)} ! vfhipvideofilter sepia=0.4 ! fakesink")
    ok(f"{SRC} ! {caps('BGRA', 320, 240)} ! vfhipconvertscale async-depth=1 ! {hipcaps('NV12', 160, 120)} ! vfhiptransform method=vertical-flip ! fakesink")
    ok(f"{SRC} ! {caps('BGRA', 320, 240)} ! vfhipconvertscale async-depth=1 ! {caps('BGRA', 320, 240)} ! fakesink")       # passthrough


def test_recurring_upstream_memory_is_page_locked_in_place(tmp_path):
    """upstream ignores the proposed pinned allocator (identity drop-allocation): its pool's system memories are
    hipHostRegister'ed once each and re-used — same bytes out, only a few registrations for 12 buffers"""
    a, b = tmp_path / "reg.raw", tmp_path / "ref.raw"
    r = gst_env.launch(f"videotestsrc num-buffers=12 pattern=ball ! {caps('NV12', 1280, 720)} ! identity drop-allocation=true ! vfhipconvertscale ! "
                       f"{caps('BGRA', 640, 360)} ! filesink location={a}", debug="vfhip:4")
    assert r.returncode == 0, r.stderr[-2000:]
    n_reg = r.stderr.count("page-locked upstream memory")
    assert 1 <= n_reg < 12, r.stderr[-2000:]
    r = gst_env.launch(f"videotestsrc num-buffers=12 pattern=ball ! {caps('NV12', 1280, 720)} ! vfhipconvertscale ! {caps('BGRA', 640, 360)} ! filesink location={b}")
    assert r.returncode == 0, r.stderr
    x, y = np.fromfile(a, np.uint8), np.fromfile(b, np.uint8)
    assert x.size == y.size == 12 * 640 * 360 * 4 and np.array_equal(x, y)


@pytest.mark.parametrize("iw,ih,ow,oh,fmt", [(1920, 1080, 640, 480, "NV12"), (1280, 720, 1920, 1080, "I420"), (720, 576, 360, 288, "NV12"), (640, 480, 320, 240, "BGRA")])
def test_pixel_parity_bicubic_with_cpu_videoscale_catrom(tmp_path, iw, ih, ow, oh, fmt):
    """method=bicubic is byte-identical to videoconvert ! videoscale method=catrom on the same frames"""
    a, b = tmp_path / "cpu.raw", tmp_path / "hip.raw"
    r = gst_env.launch(f"videotestsrc num-buffers=2 ! {caps(fmt, iw, ih)} ! tee name=t "
                       f"t. ! queue ! videoconvert ! videoscale method=catrom ! {caps('BGRA', ow, oh)} ! filesink location={a} "
                       f"t. ! queue ! vfhipconvertscale method=bicubic ! {caps('BGRA', ow, oh)} ! filesink location={b}", timeout=300)
    assert r.returncode == 0, r.stderr
    x, y = np.fromfile(a, np.uint8), np.fromfile(b, np.uint8)
    assert x.size == y.size == 2 * ow * oh * 4
    assert np.array_equal(x, y), f"max diff {np.abs(x.astype(int) - y.astype(int)).max()}, {(x != y).sum()} bytes differ"


# ---- overlay (reference tests/test-overlay.sh shapes + pixel parity with the oracle) -----------------------------------
def _write_logo(path, w=48, h=32):
    import png_util
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.zeros((h, w, 4), np.uint8)
    img[..., 0], img[..., 1], img[..., 2] = xx * 255 // (w - 1), yy * 255 // (h - 1), 200
    img[..., 3] = np.clip(255 - 8 * np.hypot(xx - w / 2, yy - h / 2), 0, 255)
    png_util.write_png(path, img, 6, 8, filters=[4])
    pre = img.copy()
    pre[..., :3] = (img[..., :3].astype(np.uint32) * img[..., 3:4] + 127) // 255
    return pre


@pytest.mark.parametrize("fmt", ["BGRA", "RGBA", "NV12", "I420"])
def test_overlay_pipelines(tmp_path, fmt):
    logo = tmp_path / "logo.png"
    _write_logo(logo)
    ok(f"{SRC} ! {caps(fmt, 320, 240)} ! vfhipoverlay location={logo} x=20 y=10 ! fakesink")
    ok(f"{SRC} ! {caps(fmt, 320, 240)} ! vfhipoverlay location={logo} relative-x=0.8 relative-y=0.05 width=60 height=40 alpha=0.5 ! fakesink")
    ok(f"{SRC} ! {caps(fmt, 320, 240)} ! vfhipoverlay ! fakesink")                                   # no image: passthrough
    ok(f"{SRC} ! {caps(fmt, 320, 240)} ! vfhipoverlay location={tmp_path / 'missing.png'} ! fakesink")   # warning, passthrough
    try:
        from PIL import Image
    except ImportError:
        return
    jpg = tmp_path / "logo.jpg"                                                                       # a JPEG logo, as the reference's ImageIO loader takes
    Image.fromarray((np.arange(48 * 64 * 3) % 251).astype(np.uint8).reshape(48, 64, 3)).save(jpg, quality=80)
    ok(f"{SRC} ! {caps(fmt, 320, 240)} ! vfhipoverlay location={jpg} x=100 y=80 alpha=0.7 ! fakesink")


def test_overlay_pixels_match_oracle(tmp_path):
    import oracle_lib
    mr = oracle_lib.load_metalref()
    logo = tmp_path / "logo.png"
    pre = _write_logo(logo)
    a, o = tmp_path / "in.raw", tmp_path / "out.raw"
    r = gst_env.launch(f"videotestsrc num-buffers=2 pattern=ball ! {caps('NV12', 320, 240)} ! tee name=t t. ! queue ! filesink location={a} "
                       f"t. ! queue ! vfhipoverlay location={logo} relative-x=0.5 y=30 width=96 alpha=0.75 ! {caps('NV12', 320, 240)} ! filesink location={o}")
    assert r.returncode == 0, r.stderr
    n = 320 * 240 * 3 // 2
    fa, fo = np.fromfile(a, np.uint8), np.fromfile(o, np.uint8)
    assert fa.size == fo.size == 2 * n
    for k in range(2):
        want = mr.overlay("NV12", 320, 240, fa[k * n:(k + 1) * n], "NV12", pre, x=160.0, y=30.0, width=96.0, height=0.0, alpha=0.75)
        got = fo[k * n:(k + 1) * n]
        d = np.abs(want.astype(int) - got.astype(int))
        assert d.max() <= 1 and (d > 0).mean() < 0.02, f"frame {k}: max {d.max()}, {(d > 0).sum()} bytes differ"


@pytest.mark.parametrize("fmt", ["UYVY", "YUY2"])
def test_pixel_parity_packed_inputs_with_cpu_elements(tmp_path, fmt):
    a, b = tmp_path / "cpu.raw", tmp_path / "hip.raw"
    r = gst_env.launch(f"videotestsrc num-buffers=2 ! {caps(fmt, 1280, 720)} ! tee name=t "
                       f"t. ! queue ! videoconvert ! videoscale ! {caps('BGRA', 640, 480)} ! filesink location={a} "
                       f"t. ! queue ! vfhipconvertscale ! {caps('BGRA', 640, 480)} ! filesink location={b}", timeout=300)
    assert r.returncode == 0, r.stderr
    x, y = np.fromfile(a, np.uint8), np.fromfile(b, np.uint8)
    assert x.size == y.size == 2 * 640 * 480 * 4 and np.array_equal(x, y)


def test_no_per_frame_leak_in_the_element_shells():
    """peak RSS of a 4-element pipeline (async convertscale, device buffers in between, overlay-less) does not grow with the
    number of frames: maps, buffer refs, pending frames and registrations are all released per frame"""
    import resource

    def peak_kb(n):
        """peak RSS over all children waited for so far (a running maximum: the short run goes first)"""
        cmd = (f"videotestsrc num-buffers={n} ! {caps('NV12', 640, 360)} ! identity drop-allocation=true ! vfhipdeinterlace method=greedyh ! "
               f"vfhipconvertscale async-depth=1 ! {caps('BGRA', 320, 180)} ! vfhipvideofilter brightness=0.1 ! vfhiptransform method=vertical-flip ! fakesink")
        r = gst_env.launch(cmd, timeout=300)
        assert r.returncode == 0, r.stderr[-1500:]
        return resource.getrusage(resource.RUSAGE_CHILDREN).ru_maxrss

    a, b = peak_kb(150), peak_kb(1500)
    assert b < a * 1.08 + 20000, f"peak RSS grew from {a} kB (150 frames) to {b} kB (1500 frames)"


@pytest.mark.parametrize("element,cin,cout", [
    ("vfhipvideofilter brightness=0.1 sharpness=0.4 noise=0.2", ("BGRA", 320, 240), ("BGRA", 320, 240)),
    ("vfhipvideofilter gamma=1.4 sepia=0.3", ("NV12", 320, 240), ("NV12", 320, 240)),
    ("vfhiptransform method=clockwise crop-left=8", ("I420", 320, 240), ("I420", 320, 240)),
    ("vfhipdeinterlace method=greedyh field-layout=top-field-first motion-threshold=0.05", ("NV12", 320, 240), ("NV12", 320, 240)),
    ("vfhipdeinterlace method=weave", ("BGRA", 320, 240), ("BGRA", 320, 240)),
    ("vfhipoverlay location=LOGO relative-x=0.5 y=20 alpha=0.7", ("BGRA", 320, 240), ("BGRA", 320, 240)),
])
@pytest.mark.parametrize("n", [1, 5])
def test_async_depth_other_elements(tmp_path, element, cin, cout, n):
    """async-depth=1 on the GstVideoFilter-based elements: same frames, same order (noise uses the per-frame counter), none
    lost at EOS"""
    logo = tmp_path / "logo.png"
    _write_logo(logo)
    element = element.replace("LOGO", str(logo))
    outs = []
    for depth in (1, 0):
        path = tmp_path / f"d{depth}.raw"
        r = gst_env.launch(f"videotestsrc num-buffers={n} pattern=ball ! {caps(*cin)} ! {element} async-depth={depth} ! {caps(*cout)} ! filesink location={path}")
        assert r.returncode == 0, r.stderr
        outs.append(np.fromfile(path, np.uint8))
    assert outs[0].size == outs[1].size and outs[0].size > 0 and outs[0].size % n == 0
    assert np.array_equal(outs[0], outs[1])


def test_compositor_output_in_hip_memory(tmp_path):
    """the composited frame handed to a downstream vfhip element as memory:HIPMemory gives the same bytes as through system memory"""
    a, b = tmp_path / "hip.raw", tmp_path / "sys.raw"
    for path, mid in ((a, hipcaps('BGRA', 320, 240)), (b, caps('BGRA', 320, 240))):
        r = gst_env.launch(f"vfhipcompositor name=comp background=white sink_1::xpos=40 sink_1::ypos=30 sink_1::alpha=0.6 ! {mid} ! vfhipvideofilter sepia=0.5 ! "
                           f"{caps('BGRA', 320, 240)} ! filesink location={path} "
                           f"videotestsrc num-buffers=3 ! {caps('BGRA', 320, 240)} ! comp. videotestsrc num-buffers=3 pattern=ball ! {caps('NV12', 160, 120)} ! comp.")
        assert r.returncode == 0, r.stderr
    x, y = np.fromfile(a, np.uint8), np.fromfile(b, np.uint8)
    assert x.size == y.size == 3 * 320 * 240 * 4 and np.array_equal(x, y)
    ok(f"vfhipcompositor name=comp ! fakesink videotestsrc num-buffers=3 ! {caps('BGRA', 320, 240)} ! comp.")         # HIPMemory straight into fakesink


@pytest.mark.parametrize("ifmt,iw,ih,ofmt,ow,oh", [("NV12", 1280, 720, "UYVY", 640, 360), ("BGRA", 640, 480, "YUY2", 320, 200), ("I420", 640, 360, "UYVY", 1280, 720),
                                                   ("UYVY", 1280, 720, "I420", 640, 480), ("YUY2", 720, 576, "NV12", 1024, 576), ("UYVY", 720, 480, "YUY2", 360, 240),
                                                   ("NV12", 1280, 720, "NV12", 640, 360), ("BGRA", 1920, 1080, "I420", 640, 480), ("I420", 1280, 720, "NV12", 854, 480)])
def test_pixel_parity_yuv_outputs_with_cpu_elements(tmp_path, ifmt, iw, ih, ofmt, ow, oh):
    """packed 4:2:2 / 4:2:0 outputs as real pipelines, caps WITHOUT colorimetry across an HD -> SD size change (GStreamer's
    by-height defaults come from the input height there): vfhipconvertscale == videoconvert ! videoscale, byte for byte"""
    a, b = tmp_path / "cpu.raw", tmp_path / "hip.raw"
    r = gst_env.launch(f"videotestsrc num-buffers=2 ! {caps(ifmt, iw, ih)} ! tee name=t "
                       f"t. ! queue ! videoconvert ! videoscale ! {caps(ofmt, ow, oh)} ! filesink location={a} "
                       f"t. ! queue ! vfhipconvertscale ! {caps(ofmt, ow, oh)} ! filesink location={b}", timeout=300)
    assert r.returncode == 0, r.stderr
    x, y = np.fromfile(a, np.uint8), np.fromfile(b, np.uint8)
    assert x.size == y.size and x.size > 0
    assert np.array_equal(x, y), f"max diff {np.abs(x.astype(int) - y.astype(int)).max()}, {(x != y).sum()} bytes differ"


@pytest.mark.parametrize("ifmt,iw,ih,ofmt,ow,oh,method,gst", [("NV12", 1280, 720, "NV12", 854, 480, "bicubic", "catrom"), ("I420", 640, 360, "UYVY", 1280, 720, "bicubic", "catrom"),
                                                              ("BGRA", 1280, 720, "I420", 640, 480, "bicubic", "catrom"), ("NV12", 1280, 720, "YUY2", 640, 360, "nearest", "nearest-neighbour")])
def test_pixel_parity_yuv_outputs_other_methods(tmp_path, ifmt, iw, ih, ofmt, ow, oh, method, gst):
    """method=bicubic / nearest with YUV outputs, as real pipelines against the CPU elements"""
    a, b = tmp_path / "cpu.raw", tmp_path / "hip.raw"
    r = gst_env.launch(f"videotestsrc num-buffers=2 ! {caps(ifmt, iw, ih)} ! tee name=t "
                       f"t. ! queue ! videoconvert ! videoscale method={gst} ! {caps(ofmt, ow, oh)} ! filesink location={a} "
                       f"t. ! queue ! vfhipconvertscale method={method} ! {caps(ofmt, ow, oh)} ! filesink location={b}", timeout=300)
    assert r.returncode == 0, r.stderr
    x, y = np.fromfile(a, np.uint8), np.fromfile(b, np.uint8)
    assert x.size == y.size and x.size > 0
    assert np.array_equal(x, y), f"max diff {np.abs(x.astype(int) - y.astype(int)).max()}, {(x != y).sum()} bytes differ"


@pytest.mark.parametrize("n", [1, 6])
@pytest.mark.parametrize("ofmt", ["BGRA", "NV12"])
def test_compositor_async_depth_same_frames(tmp_path, n, ofmt):
    """vfhipcompositor async-depth=1 (submit this composite, then complete the previous one): the same frames in the same
    order as the synchronous element, none lost at EOS"""
    outs = []
    for depth in (1, 0):
        path = tmp_path / f"c{depth}.raw"
        r = gst_env.launch(f"vfhipcompositor name=c background=checker async-depth={depth} sink_0::alpha=0.8 sink_1::xpos=100 sink_1::ypos=40 sink_1::width=200 sink_1::height=120 "
                           f"sink_1::operator=add ! {caps(ofmt, 480, 270)} ! filesink location={path} "
                           f"videotestsrc num-buffers={n} pattern=ball ! {caps('BGRA', 320, 240)} ! c.sink_0 "
                           f"videotestsrc num-buffers={n} pattern=smpte ! {caps('NV12', 160, 120)} ! c.sink_1", timeout=300)
        assert r.returncode == 0, r.stderr
        outs.append(np.fromfile(path, np.uint8))
    assert outs[0].size == outs[1].size and outs[0].size > 0 and outs[0].size % n == 0
    assert np.array_equal(outs[0], outs[1])


def test_compositor_skips_obscured_pads(tmp_path):
    """a pad completely behind a later opaque pad is neither uploaded nor drawn (reference pad_obscures_rectangle,
    gstvfmetalcompositor.m:329-358) and the frame is the same as without it; an `add` pad on top obscures nothing"""
    def run(top_op, with_lower, debug=None):
        path = tmp_path / f"o_{top_op}_{int(with_lower)}.raw"
        lower = f"videotestsrc num-buffers=2 pattern=ball ! {caps('BGRA', 160, 120)} ! c.sink_0 " if with_lower else ""
        pads = "sink_0::xpos=50 sink_0::ypos=40 sink_1::operator=" + top_op if with_lower else "sink_0::operator=" + top_op
        r = gst_env.launch(f"vfhipcompositor name=c background=black {pads} ! {caps('BGRA', 320, 240)} ! filesink location={path} {lower}"
                           f"videotestsrc num-buffers=2 pattern=smpte ! {caps('NV12', 320, 240)} ! c.sink_{1 if with_lower else 0}", timeout=60, debug=debug)
        assert r.returncode == 0, r.stderr
        return np.fromfile(path, np.uint8), r.stderr
    both, log = run("over", True, debug="vfhip*:7")
    only, _ = run("over", False)
    assert both.size == only.size == 2 * 320 * 240 * 4 and np.array_equal(both, only)
    assert "obscured by a later opaque pad" in log
    added, log = run("add", True, debug="vfhip*:7")
    assert "obscured by a later opaque pad" not in log and not np.array_equal(added, only)


def test_compositor_forwards_pointer_events_to_the_pads_under_them(tmp_path):
    """navigation (mouse) events sent upstream reach the sink pads whose picture is under the pointer, in that input's own
    pixel coordinates (reference _src_event, gstvfmetalcompositor.m:704-787)"""
    import os
    import subprocess
    exe = tmp_path / "nav_probe"
    inc = ["-I/opt/conda/include/gstreamer-1.0", "-I/opt/conda/include/glib-2.0", "-I/opt/conda/lib/glib-2.0/include"]
    lib = ["-L/opt/conda/lib", "-lgstreamer-1.0", "-lgobject-2.0", "-lglib-2.0", "-Wl,-rpath,/opt/conda/lib"]
    subprocess.check_call(["gcc", "-O1", "-o", str(exe), os.path.join(os.path.dirname(__file__), "nav_probe.c")] + inc + lib)

    def seen(x, y):
        r = subprocess.run([str(exe), str(x), str(y)], env=gst_env.env(), capture_output=True, text=True, timeout=90)
        assert r.returncode == 0, r.stderr
        return {ln.split()[0]: (float(ln.split()[1]), float(ln.split()[2])) for ln in r.stdout.splitlines() if ln and ln[0] in "ab"}
    # output = bounding box 360 x 240; pad b is 80x120 shown as 160x60 at (200, 100)
    assert seen(10, 20) == {"a": (10.0, 20.0)}                                   # only the full-size pad
    both = seen(240, 130)                                                          # inside both pictures
    assert both["a"] == (240.0, 130.0) and both["b"] == ((240 - 200) * 80 / 160, (130 - 100) * 120 / 60)
    assert seen(340, 120) == {"b": ((340 - 200) * 80 / 160, (120 - 100) * 120 / 60)}   # right of pad a's 320 columns
    assert seen(340, 200) == {}                                                   # background only


def test_compositor_mixed_frame_rates(tmp_path):
    """a 30 fps and a 15 fps input into one 30 fps output (the reference is a GstVideoAggregator: per-pad buffer selection by
    running time, gstvfmetalcompositor.m:171-174; its tests/test-compositor.sh mixes sources freely): the output runs at the
    faster rate for as long as the inputs last, the fast pad contributes every one of its frames in order, the slow pad's
    frames are each shown for two consecutive output frames (one either way at the 1 ns rounding of the frame times)."""
    a, b, o = tmp_path / "a.raw", tmp_path / "b.raw", tmp_path / "o.raw"
    r = gst_env.launch(f"vfhipcompositor name=c background=black sink_1::xpos=320 ! {caps('BGRA', 640, 240)},framerate=30/1 ! filesink location={o} "
                       f"videotestsrc num-buffers=12 pattern=ball ! {caps('BGRA', 320, 240)},framerate=30/1 ! tee name=ta ta. ! queue ! c.sink_0 ta. ! queue ! filesink location={a} "
                       f"videotestsrc num-buffers=6 pattern=ball ! {caps('BGRA', 320, 240)},framerate=15/1 ! tee name=tb tb. ! queue ! c.sink_1 tb. ! queue ! filesink location={b}", timeout=120)
    assert r.returncode == 0, r.stderr
    fa, fb, fo = np.fromfile(a, np.uint8).reshape(-1, 240, 320, 4), np.fromfile(b, np.uint8).reshape(-1, 240, 320, 4), np.fromfile(o, np.uint8).reshape(-1, 240, 640, 4)
    assert len(fa) == 12 and len(fb) == 6
    assert len(fo) == 12, f"{len(fo)} output frames: 0.4 s of input at 30 fps is 12"
    assert all(not np.array_equal(fb[j], fb[j + 1]) for j in range(5)), "the slow source must change from frame to frame for this test to see repeats"
    shown = []
    for k in range(12):
        assert np.array_equal(fo[k][:, :320], fa[k]), f"output frame {k}: the 30 fps pad must show its frame {k}"
        match = [j for j in range(6) if np.array_equal(fo[k][:, 320:], fb[j])]
        assert match, f"output frame {k}: the right half is none of the 15 fps pad's frames"
        assert match[0] in (k // 2, (k + 1) // 2), f"output frame {k} shows slow frame {match[0]}"
        shown.append(match[0])
    assert shown == sorted(shown) and set(shown) == set(range(6)), shown        # in order, none skipped, each repeated
    assert max(shown.count(j) for j in range(6)) <= 3


def test_compositor_drops_frames_of_a_faster_pad_and_repeats_after_eos(tmp_path):
    """a 60 fps pad into a 30 fps output is decimated (old buffers dropped, never queued up: the output stays 0.2 s long), and a
    pad with repeat-after-eos keeps its last frame on screen while the other pad runs on"""
    o = tmp_path / "o.raw"
    r = gst_env.launch(f"vfhipcompositor name=c background=black sink_1::xpos=160 ! {caps('BGRA', 320, 120)},framerate=30/1 ! filesink location={o} "
                       f"videotestsrc num-buffers=6 pattern=ball ! {caps('BGRA', 160, 120)},framerate=30/1 ! c.sink_0 "
                       f"videotestsrc num-buffers=12 pattern=ball ! {caps('BGRA', 160, 120)},framerate=60/1 ! c.sink_1", timeout=120)
    assert r.returncode == 0, r.stderr
    # 0.2 s of input: 6 output frames, or 7 when the last 60 fps buffer (which ends exactly where output frame 6 starts: GstVideoAggregator
    # takes `end >= out_start`) is still queued — never 12: the fast pad is decimated, not queued up
    assert np.fromfile(o, np.uint8).size in (6 * 320 * 120 * 4, 7 * 320 * 120 * 4)
    r = gst_env.launch(f"vfhipcompositor name=c background=black sink_0::repeat-after-eos=true sink_1::xpos=160 ! {caps('BGRA', 320, 120)},framerate=30/1 ! filesink location={o} "
                       f"videotestsrc num-buffers=2 pattern=smpte ! {caps('BGRA', 160, 120)},framerate=30/1 ! c.sink_0 "
                       f"videotestsrc num-buffers=6 pattern=ball ! {caps('BGRA', 160, 120)},framerate=30/1 ! c.sink_1", timeout=120)
    assert r.returncode == 0, r.stderr
    fo = np.fromfile(o, np.uint8).reshape(-1, 120, 320, 4)
    assert len(fo) == 6
    assert all(np.array_equal(fo[k][:, :160], fo[1][:, :160]) for k in range(1, 6)) and fo[5][:, :160].any()      # the ended pad's last frame stays


def test_compositor_qos_skips_late_frames(tmp_path):
    """GstVideoAggregator's QoS, which the reference inherits: a sink that reports lateness (sync + qos behind an element that takes 60 ms per 33 ms frame)
    makes the compositor skip output frames that are already late — they are not composited, their time passes, a QoS message is posted — instead of
    compositing every frame for a sink that throws it away.  Without the QOS events (qos=false) every frame is composited."""
    pipe = (f"vfhipcompositor name=c background=black ! {caps('BGRA', 320, 240)},framerate=30/1 ! identity sleep-time=60000 ! fakesink sync=true qos={{q}} "
            f"videotestsrc num-buffers=45 pattern=ball ! {caps('BGRA', 320, 240)},framerate=30/1 ! c.sink_0")
    r = gst_env.launch(pipe.format(q="true"), timeout=180, debug="vfhipcompositor:5")
    assert r.returncode == 0, r.stderr[-2000:]
    late = [ln for ln in r.stderr.splitlines() if "not composited" in ln]
    assert 3 <= len(late) < 45, (len(late), r.stderr[-1500:])
    r = gst_env.launch(pipe.format(q="false"), timeout=180, debug="vfhipcompositor:5")
    assert r.returncode == 0 and not [ln for ln in r.stderr.splitlines() if "not composited" in ln]


def test_compositor_max_last_buffer_repeat_leaves_covered_frames_alone(tmp_path):
    """max-last-buffer-repeat (GstVideoAggregatorPad's) caps how long a pad that is NOT at EOS shows a buffer past the buffer's END; a slower pad whose
    buffers still cover the output frames is not touched by it, even at 0"""
    o, o2 = tmp_path / "a.raw", tmp_path / "b.raw"
    pipe = (f"vfhipcompositor name=c background=black sink_1::xpos=160 {{extra}} ! {caps('BGRA', 320, 120)},framerate=30/1 ! filesink location={{out}} "
            f"videotestsrc num-buffers=6 pattern=ball ! {caps('BGRA', 160, 120)},framerate=30/1 ! c.sink_0 "
            f"videotestsrc num-buffers=3 pattern=smpte ! {caps('BGRA', 160, 120)},framerate=15/1 ! c.sink_1")
    r = gst_env.launch(pipe.format(extra="", out=o), timeout=120)
    assert r.returncode == 0, r.stderr
    r = gst_env.launch(pipe.format(extra="sink_1::max-last-buffer-repeat=0 sink_0::max-last-buffer-repeat=0", out=o2), timeout=120)
    assert r.returncode == 0, r.stderr
    a, b = np.fromfile(o, np.uint8), np.fromfile(o2, np.uint8)
    assert a.size >= 6 * 320 * 120 * 4 and np.array_equal(a, b)


@pytest.mark.parametrize("element,chain", [("vfhipconvertscale", "vfhipconvertscale ! video/x-raw,format=BGRA,width=160,height=120"), ("vfhipvideofilter", "vfhipvideofilter brightness=0.1"),
                                           ("vfhipdeinterlace", "vfhipdeinterlace"), ("vfhiptransform", "vfhiptransform method=clockwise"), ("vfhipoverlay", "vfhipoverlay"),
                                           ("vfhipcompositor", "vfhipcompositor")])
def test_every_element_has_its_own_debug_category(element, chain):
    """GST_DEBUG=<element>:6 selects that element's log lines only (the reference registers one category per element, e.g.
    convertscale/gstvfmetalconvertscale.m:538-539); the shared helpers stay on `vfhip`"""
    r = gst_env.launch(f"{SRC} ! {caps('NV12', 320, 240)} ! {chain} ! fakesink", debug=f"{element}:6")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stderr.splitlines() if " vfhip" in ln]
    assert lines, "no log line at level 6 from the element"
    assert all(f" {element} " in ln for ln in lines), [ln for ln in lines if f" {element} " not in ln][:3]
