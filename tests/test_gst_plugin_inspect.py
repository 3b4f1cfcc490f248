"""The GStreamer plugin (plain C element shells over libvfhip) keeps the reference's element API: same base classes,
pad templates, property names / nicks / ranges / defaults.  Mirrors the `check_inspect` part of the
reference's smoke scripts (tests/test-videofilter.sh:69-97, tests/test-convertscale.sh:27-39) with vfhip* names.
gst-inspect instantiates the elements but creates no renderer, so this runs without a GPU."""
import re

import pytest

import gst_env

pytestmark = pytest.mark.skipif(not gst_env.available(), reason="GStreamer 1.14 (/opt/conda) or libgstvfhip.so not present")


def props(text):
    body = text.split("Element Properties:")[1]
    return set(re.findall(r"^  ([a-z][a-z0-9-]*) *:", body, flags=re.M))


def squeeze(text):
    return re.sub(r"\s+", " ", text)


def test_plugin_registers_elements():
    r = gst_env.inspect("vfhip")
    assert r.returncode == 0, r.stdout + r.stderr
    for e in ("vfhipconvertscale", "vfhipvideofilter", "vfhipdeinterlace", "vfhiptransform", "vfhipcompositor", "vfhipoverlay"):
        assert e in r.stdout


def test_convertscale_api():
    t = gst_env.inspect("vfhipconvertscale").stdout
    assert {"method", "add-borders", "border-color", "device-id", "numerics", "async-depth"} <= props(t)
    assert "GstBaseTransform" in t and "bilinear" in t and "nearest" in t
    # both pads: the six formats in system memory and as memory:HIPMemory (device-resident buffers between vfhip elements)
    assert t.count("(string)BGRA, (string)RGBA, (string)NV12, (string)I420, (string)UYVY, (string)YUY2") == 4
    assert t.count("video/x-raw(memory:HIPMemory)") == 2
    assert "Default: 4278190080" in t                       # border-color 0xFF000000


def test_videofilter_api():
    t = gst_env.inspect("vfhipvideofilter").stdout
    want = {"brightness", "contrast", "saturation", "hue", "gamma", "sharpness", "sepia", "invert", "noise", "vignette",
            "chroma-key-enabled", "chroma-key-color", "chroma-key-tolerance", "chroma-key-smoothness", "lut-file"}
    assert want <= props(t) and len(want) == 15 and "async-depth" in props(t)
    assert "GstVideoFilter" in t
    assert t.count("(string)BGRA, (string)RGBA, (string)NV12, (string)I420 }") == 4 and t.count("video/x-raw(memory:HIPMemory)") == 2
    assert "Range: 0.01 - 10 Default: 1 " in squeeze(t)      # gamma
    assert "Default: 4278255360" in t                       # chroma-key-color 0xFF00FF00


def test_deinterlace_api():
    t = gst_env.inspect("vfhipdeinterlace").stdout
    assert {"method", "field-layout", "motion-threshold"} <= props(t)
    for nick in ("bob", "weave", "linear", "greedyh", "auto", "top-field-first", "bottom-field-first"):
        assert nick in t
    assert "Range: 0 - 1 Default: 0.1" in squeeze(t)


def test_transform_api():
    t = gst_env.inspect("vfhiptransform").stdout
    assert {"method", "crop-top", "crop-bottom", "crop-left", "crop-right"} <= props(t)
    for nick in ("none", "clockwise", "rotate-180", "counterclockwise", "horizontal-flip", "vertical-flip", "upper-left-diagonal", "upper-right-diagonal"):
        assert nick in t
    assert "GstVideoFilter" in t


def test_compositor_api():
    """reference tests/test-compositor.sh:60-88: element + pad property surface (pad properties show on a requested pad)"""
    t = gst_env.inspect("vfhipcompositor").stdout
    assert {"background", "zero-size-is-unscaled", "ignore-inactive-pads", "device-id", "async-depth"} <= props(t)
    for nick in ("checker", "black", "white", "transparent"):
        assert nick in t
    assert "GstAggregator" in t and "GstChildProxy" in t and "sink_%u" in t and "On request" in t
    assert "primary + 2" in t
    assert t.count("(string)BGRA, (string)RGBA, (string)NV12, (string)I420 }") == 4          # src and sink_%u, each in both memories
    src = open(gst_env.PLUGIN_DIR + "/gstvfhipcompositor.c").read()
    # (... and GstVideoAggregatorPad's own, which the reference's pads inherit)
    for prop in ("xpos", "ypos", "width", "height", "alpha", "operator", "sizing-policy", "zorder", "repeat-after-eos", "max-last-buffer-repeat"):
        assert f'("{prop}"' in src


def test_overlay_api():
    """reference overlay/gstvfmetaloverlay.m:375-420: names, ranges, defaults"""
    t = gst_env.inspect("vfhipoverlay").stdout
    assert {"location", "x", "y", "width", "height", "alpha", "relative-x", "relative-y", "device-id"} <= props(t)
    assert "GstVideoFilter" in t
    sq = squeeze(t)
    assert "Range: -1 - 1 Default: -1" in sq and "Range: 0 - 1 Default: 1" in sq and "Range: 0 - 2147483647 Default: 0" in sq
    assert t.count("(string)BGRA, (string)RGBA, (string)NV12, (string)I420 }") == 4
