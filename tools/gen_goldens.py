#!/usr/bin/env python3
"""tools/gen_goldens.py — regenerate tests/golden/*.npz from the REAL GStreamer 1.14.0 elements.

Runs only in the build container (GStreamer 1.14.0 under /opt/conda); never on the
GPU box.  It is the script that made the committed fixtures: every expected output below is what
`videoconvert ! videoscale` (the oracle BASELINE.json's north_star names) produced for the stored
input.  No reference (visioforge/gstreamer-metal) code is involved: that plugin cannot be built on
Linux and its tests pin no pixel values.

    python tools/gen_goldens.py            # rewrites tests/golden/convertscale_gst114*.npz
"""
import hashlib, json, os, subprocess, sys, tempfile
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
ENV = dict(os.environ)
ENV.update(PATH="/opt/conda/bin:" + ENV["PATH"], GST_PLUGIN_SYSTEM_PATH="/opt/conda/lib/gstreamer-1.0",
           GST_PLUGIN_SCANNER="/opt/conda/libexec/gstreamer-1.0/gst-plugin-scanner",
           GST_REGISTRY="/tmp/gst-registry.bin", LD_LIBRARY_PATH="/opt/conda/lib")
INC = ["-I/opt/conda/include/gstreamer-1.0", "-I/opt/conda/include/glib-2.0", "-I/opt/conda/lib/glib-2.0/include"]
LIB = ["-L/opt/conda/lib", "-lgstreamer-1.0", "-lgobject-2.0", "-lglib-2.0", "-Wl,-rpath,/opt/conda/lib"]


def build_helper(tmp):
    exe = os.path.join(tmp, "gst114_run")
    subprocess.check_call(["gcc", "-O1", "-o", exe, os.path.join(ROOT, "tools", "gst114_run.c")] + INC + LIB)
    return exe


def gst_run(exe, tmp, raw, frame_bytes, incaps, middle, outcaps):
    i, o = os.path.join(tmp, "in.raw"), os.path.join(tmp, "out.raw")
    with open(i, "wb") as f:
        f.write(raw)
    r = subprocess.run([exe, i, str(frame_bytes), incaps, middle, outcaps, o], env=ENV, capture_output=True, text=True)
    if r.returncode:
        raise RuntimeError(r.stderr)
    with open(o, "rb") as f:
        return f.read()


def videotestsrc(tmp, fmt, w, h, pattern="smpte"):
    o = os.path.join(tmp, "vts.raw")
    cmd = (f"gst-launch-1.0 -q videotestsrc num-buffers=1 pattern={pattern} ! video/x-raw,format={fmt},width={w},height={h} "
           f"! filesink location={o}")
    subprocess.check_call(cmd, shell=True, env=ENV)
    with open(o, "rb") as f:
        return f.read()


def r4(x):
    return (x + 3) // 4 * 4


def nv12_layout(w, h):
    """GstVideoInfo default layout: strides rounded to 4, UV plane at stride*ROUND_UP_2(h)."""
    ys = r4(w); hp = (h + 1) // 2 * 2
    return ys, ys * hp, ys, ys * hp + ys * (hp // 2)          # ystride, uv offset, uvstride, size


def i420_layout(w, h):
    ys = r4(w); hp = (h + 1) // 2 * 2; cs = r4((w + 1) // 2)
    uo = ys * hp; vo = uo + cs * (hp // 2)
    return ys, uo, vo, cs, vo + cs * (hp // 2)


def rand_frame(rng, fmt, w, h):
    if fmt == "NV12":
        ys, uo, us, size = nv12_layout(w, h)
    else:
        ys, uo, vo, cs, size = i420_layout(w, h)
    return rng.integers(0, 256, size, dtype=np.uint8).tobytes()


def main():
    os.makedirs(GOLD, exist_ok=True)
    cases, arrays = [], {}
    with tempfile.TemporaryDirectory() as tmp:
        exe = build_helper(tmp)

        def add(name, fmt, w, h, raw, col, site, method, ofmt, ow, oh):
            caps = f"video/x-raw,format={fmt},width={w},height={h},framerate=1/1"
            if col:
                caps += f",colorimetry={col}"
            if site:
                caps += f",chroma-site={site}"
            m = "nearest-neighbour" if method == "nearest" else "bilinear"
            out = gst_run(exe, tmp, raw, len(raw), caps, f"videoconvert ! videoscale method={m}",
                          f"video/x-raw,format={ofmt},width={ow},height={oh}")
            assert len(out) == ow * oh * 4, (name, len(out))
            arrays[name + "_in"] = np.frombuffer(raw, np.uint8)
            arrays[name + "_out"] = np.frombuffer(out, np.uint8)
            cases.append(dict(name=name, in_format=fmt, w=w, h=h, colorimetry=col, chroma_site=site, method=method,
                              out_format=ofmt, ow=ow, oh=oh, in_sha256=hashlib.sha256(raw).hexdigest(),
                              out_sha256=hashlib.sha256(out).hexdigest()))
            print(name, cases[-1]["out_sha256"][:16])

        # BASELINE configs 0 and 1: videotestsrc frame 0, caps carry no colorimetry -> GStreamer's by-height default
        add("c1_vts_1080_to_640x480", "NV12", 1920, 1080, videotestsrc(tmp, "NV12", 1920, 1080), None, None, "bilinear", "BGRA", 640, 480)
        add("c2_vts_2160_to_1080", "NV12", 3840, 2160, videotestsrc(tmp, "NV12", 3840, 2160), None, None, "bilinear", "BGRA", 1920, 1080)

        rng = np.random.default_rng(20261004)
        cols, sites = ["bt601", "bt709", "bt2020"], ["jpeg", "mpeg2"]
        fixed = [(64, 36, 32, 18), (128, 72, 50, 30), (64, 36, 64, 36), (64, 36, 100, 50), (48, 40, 20, 37), (48, 40, 96, 38),
                 (2, 2, 2, 2), (3, 3, 7, 5), (16, 16, 1, 1), (33, 17, 16, 8), (200, 8, 100, 4), (8, 200, 4, 100)]
        t = 0
        for fmt in ["NV12", "I420"]:
            for (w, h, ow, oh) in fixed + [tuple(int(v) for v in rng.integers(2, 97, 4)) for _ in range(18)]:
                col, site = cols[t % 3], sites[(t // 3) % 2]
                method = "nearest" if t % 7 == 3 else "bilinear"
                ofmt = "RGBA" if t % 5 == 1 else "BGRA"
                add(f"{fmt.lower()}_{t:03d}_{w}x{h}_to_{ow}x{oh}", fmt, w, h, rand_frame(rng, fmt, w, h), col, site, method, ofmt, ow, oh)
                t += 1
    arrays["manifest"] = np.frombuffer(json.dumps(cases).encode(), np.uint8)
    np.savez_compressed(os.path.join(GOLD, "convertscale_gst114.npz"), **arrays)
    print("wrote", len(cases), "cases")
    gen_yuv_outputs()


def rgb_layout_size(w, h):
    return 4 * w * h


def gen_bicubic():
    """`videoconvert ! videoscale method=catrom` -> tests/golden/convertscale_gst114_bicubic.npz (RGB outputs)"""
    cases, arrays = [], {}
    with tempfile.TemporaryDirectory() as tmp:
        exe = build_helper(tmp)
        rng = np.random.default_rng(20261005)
        cols, sites = ["bt601", "bt709", "bt2020"], ["jpeg", "mpeg2"]
        sizes = [(64, 36, 32, 18), (64, 36, 128, 72), (48, 40, 20, 37), (48, 40, 96, 38), (33, 17, 16, 8), (200, 8, 100, 4), (8, 200, 4, 100),
                 (64, 16, 32, 8), (64, 18, 32, 9), (17, 13, 11, 29), (96, 54, 96, 20), (96, 54, 31, 54), (3, 3, 7, 5), (40, 30, 9, 7), (16, 16, 1, 1),
                 (300, 12, 100, 12), (195, 9, 65, 21)]    # 3:1 with a non-power-of-two output: exposes the order of (j + .5) / out * in
        sizes += [tuple(int(v) for v in rng.integers(2, 97, 4)) for _ in range(10)]
        t = 0
        for fmt in ["NV12", "I420", "BGRA", "RGBA"]:
            for (w, h, ow, oh) in sizes:
                if t % 2 == 1 and fmt in ("BGRA", "RGBA") and t > 8:
                    t += 1
                    continue
                col, site = cols[t % 3], sites[(t // 3) % 2]
                ofmt = "RGBA" if t % 5 == 1 else "BGRA"
                raw = rand_frame(rng, fmt, w, h) if fmt in ("NV12", "I420") else rng.integers(0, 256, 4 * w * h, dtype=np.uint8).tobytes()
                caps = f"video/x-raw,format={fmt},width={w},height={h},framerate=1/1,colorimetry={col},chroma-site={site}"
                out = gst_run(exe, tmp, raw, len(raw), caps, "videoconvert ! videoscale method=catrom", f"video/x-raw,format={ofmt},width={ow},height={oh}")
                assert len(out) == ow * oh * 4
                name = f"bc_{fmt.lower()}_{t:03d}_{w}x{h}_to_{ow}x{oh}"
                arrays[name + "_in"] = np.frombuffer(raw, np.uint8)
                arrays[name + "_out"] = np.frombuffer(out, np.uint8)
                cases.append(dict(name=name, in_format=fmt, w=w, h=h, colorimetry=col, chroma_site=site, method="bicubic", out_format=ofmt, ow=ow, oh=oh,
                                  in_sha256=hashlib.sha256(raw).hexdigest(), out_sha256=hashlib.sha256(out).hexdigest()))
                t += 1
        # the headline shape, smaller: videotestsrc 1080p -> 540p (vertical pass first)
        raw = videotestsrc(tmp, "NV12", 960, 540)
        out = gst_run(exe, tmp, raw, len(raw), "video/x-raw,format=NV12,width=960,height=540,framerate=1/1", "videoconvert ! videoscale method=catrom",
                      "video/x-raw,format=BGRA,width=480,height=270")
        arrays["bc_vts_in"], arrays["bc_vts_out"] = np.frombuffer(raw, np.uint8), np.frombuffer(out, np.uint8)
        cases.append(dict(name="bc_vts", in_format="NV12", w=960, h=540, colorimetry=None, chroma_site=None, method="bicubic", out_format="BGRA", ow=480, oh=270,
                          in_sha256=hashlib.sha256(raw).hexdigest(), out_sha256=hashlib.sha256(out).hexdigest()))
    arrays["manifest"] = np.frombuffer(json.dumps(cases).encode(), np.uint8)
    np.savez_compressed(os.path.join(GOLD, "convertscale_gst114_bicubic.npz"), **arrays)
    print("wrote", len(cases), "bicubic cases")


def gen_yuv_outputs():
    """cells whose OUTPUT is NV12 / I420 (videoconvert's RGB->YUV matrix + chroma averaging, NV12<->I420 re-packing,
    per-plane videoscale): tests/golden/convertscale_gst114_yuvout.npz"""
    cases, arrays = [], {}
    rng = np.random.default_rng(20261005)
    cols, sites = ["bt601", "bt709", "bt2020"], ["jpeg", "mpeg2"]
    fixed = [(64, 36, 64, 36), (64, 36, 32, 18), (64, 36, 32, 36), (64, 36, 32, 20), (33, 17, 33, 17), (33, 17, 64, 36), (48, 40, 20, 38),
             (200, 8, 100, 4), (2, 2, 2, 2), (3, 5, 7, 3), (66, 36, 33, 18), (16, 16, 1, 1)]
    with tempfile.TemporaryDirectory() as tmp:
        exe = build_helper(tmp)
        t = 0
        for ifmt in ["BGRA", "RGBA", "NV12", "I420"]:
            for ofmt in ["NV12", "I420"]:
                for (w, h, ow, oh) in fixed[(t % 3)::3] + [tuple(int(v) for v in rng.integers(2, 90, 4)) for _ in range(5)]:
                    col, site = cols[t % 3], sites[(t // 3) % 2]
                    size = rgb_layout_size(w, h) if ifmt in ("BGRA", "RGBA") else (nv12_layout(w, h)[3] if ifmt == "NV12" else i420_layout(w, h)[4])
                    raw = rng.integers(0, 256, size, dtype=np.uint8).tobytes()
                    incaps = f"video/x-raw,format={ifmt},width={w},height={h},framerate=1/1"
                    if ifmt in ("NV12", "I420"):
                        incaps += f",colorimetry={col},chroma-site={site}"
                    out = gst_run(exe, tmp, raw, len(raw), incaps, "videoconvert ! videoscale",
                                  f"video/x-raw,format={ofmt},width={ow},height={oh},colorimetry={col},chroma-site={site}")
                    name = f"{ifmt.lower()}_to_{ofmt.lower()}_{t:03d}_{w}x{h}_to_{ow}x{oh}"
                    arrays[name + "_in"] = np.frombuffer(raw, np.uint8)
                    arrays[name + "_out"] = np.frombuffer(out, np.uint8)
                    cases.append(dict(name=name, in_format=ifmt, w=w, h=h, colorimetry=col, chroma_site=site, method="bilinear",
                                      out_format=ofmt, ow=ow, oh=oh, in_sha256=hashlib.sha256(raw).hexdigest(),
                                      out_sha256=hashlib.sha256(out).hexdigest()))
                    t += 1
    arrays["manifest"] = np.frombuffer(json.dumps(cases).encode(), np.uint8)
    np.savez_compressed(os.path.join(GOLD, "convertscale_gst114_yuvout.npz"), **arrays)
    print("wrote", len(cases), "yuv-output cases")


def gen_packed():
    """UYVY / YUY2 inputs -> RGB outputs through `videoconvert ! videoscale` -> tests/golden/convertscale_gst114_packed.npz"""
    cases, arrays = [], {}
    with tempfile.TemporaryDirectory() as tmp:
        exe = build_helper(tmp)
        rng = np.random.default_rng(20261006)
        cols, sites = ["bt601", "bt709", "bt2020"], ["jpeg", "mpeg2"]
        sizes = [(64, 36, 64, 36), (64, 36, 32, 18), (64, 36, 100, 50), (48, 40, 20, 37), (33, 17, 33, 17), (33, 17, 16, 8), (17, 9, 40, 21), (2, 2, 2, 2),
                 (200, 8, 100, 4), (31, 30, 31, 12), (50, 20, 125, 20)]
        sizes += [tuple(int(v) for v in rng.integers(2, 97, 4)) for _ in range(6)]
        t = 0
        for fmt in ["UYVY", "YUY2"]:
            for (w, h, ow, oh) in sizes:
                col, site = cols[t % 3], sites[(t // 3) % 2]
                method = ["bilinear", "nearest", "bicubic"][t % 3] if t % 4 else "bilinear"
                if method == "bicubic" and not all(i == o or (int(np.ceil(4 * max(1.0, i / o))) <= min(i, 64)) for i, o in ((w, ow), (h, oh))):
                    method = "bilinear"
                ofmt = "RGBA" if t % 5 == 1 else "BGRA"
                stride = (2 * w + 3) // 4 * 4
                raw = rng.integers(0, 256, stride * h, dtype=np.uint8).tobytes()
                caps = f"video/x-raw,format={fmt},width={w},height={h},framerate=1/1,colorimetry={col},chroma-site={site}"
                m = {"bilinear": "bilinear", "nearest": "nearest-neighbour", "bicubic": "catrom"}[method]
                out = gst_run(exe, tmp, raw, len(raw), caps, f"videoconvert ! videoscale method={m}", f"video/x-raw,format={ofmt},width={ow},height={oh}")
                assert len(out) == ow * oh * 4
                name = f"pk_{fmt.lower()}_{t:03d}_{w}x{h}_to_{ow}x{oh}_{method}"
                arrays[name + "_in"], arrays[name + "_out"] = np.frombuffer(raw, np.uint8), np.frombuffer(out, np.uint8)
                cases.append(dict(name=name, in_format=fmt, w=w, h=h, colorimetry=col, chroma_site=site, method=method, out_format=ofmt, ow=ow, oh=oh,
                                  in_sha256=hashlib.sha256(raw).hexdigest(), out_sha256=hashlib.sha256(out).hexdigest()))
                t += 1
    arrays["manifest"] = np.frombuffer(json.dumps(cases).encode(), np.uint8)
    np.savez_compressed(os.path.join(GOLD, "convertscale_gst114_packed.npz"), **arrays)
    print("wrote", len(cases), "packed-input cases")


def gen_packed_outputs():
    """cells whose OUTPUT is UYVY / YUY2 (from all six input formats) and UYVY / YUY2 -> NV12 / I420, through
    `videoconvert ! videoscale` (bilinear) -> tests/golden/convertscale_gst114_packedout.npz"""
    cases, arrays = [], {}
    rng = np.random.default_rng(20261007)
    cols, sites = ["bt601", "bt709", "bt2020"], ["jpeg", "mpeg2"]
    fixed = [(64, 36, 64, 36), (64, 36, 32, 18), (64, 36, 100, 50), (33, 17, 33, 17), (33, 17, 16, 8), (17, 9, 40, 21), (48, 40, 20, 37),
             (200, 8, 100, 4), (2, 2, 2, 2), (31, 30, 31, 12), (50, 20, 125, 20), (35, 29, 18, 7), (3, 5, 7, 3), (16, 16, 1, 1)]
    pairs = [(i, o) for i in ["BGRA", "RGBA", "NV12", "I420", "UYVY", "YUY2"] for o in ["UYVY", "YUY2"]]
    pairs += [(i, o) for i in ["UYVY", "YUY2"] for o in ["NV12", "I420"]]
    with tempfile.TemporaryDirectory() as tmp:
        exe = build_helper(tmp)
        t = 0
        for (ifmt, ofmt) in pairs:
            for (w, h, ow, oh) in fixed[(t % 4)::4] + [tuple(int(v) for v in rng.integers(2, 90, 4)) for _ in range(4)]:
                col, site = cols[t % 3], sites[(t // 3) % 2]
                if ifmt in ("BGRA", "RGBA"):
                    size = w * h * 4
                elif ifmt == "NV12":
                    size = nv12_layout(w, h)[3]
                elif ifmt == "I420":
                    size = i420_layout(w, h)[4]
                else:
                    size = r4(2 * w) * h
                raw = rng.integers(0, 256, size, dtype=np.uint8).tobytes()
                incaps = f"video/x-raw,format={ifmt},width={w},height={h},framerate=1/1"
                if ifmt not in ("BGRA", "RGBA"):
                    incaps += f",colorimetry={col},chroma-site={site}"
                out = gst_run(exe, tmp, raw, len(raw), incaps, "videoconvert ! videoscale",
                              f"video/x-raw,format={ofmt},width={ow},height={oh},colorimetry={col},chroma-site={site}")
                name = f"{ifmt.lower()}_to_{ofmt.lower()}_{t:03d}_{w}x{h}_to_{ow}x{oh}"
                arrays[name + "_in"], arrays[name + "_out"] = np.frombuffer(raw, np.uint8), np.frombuffer(out, np.uint8)
                cases.append(dict(name=name, in_format=ifmt, w=w, h=h, colorimetry=col, chroma_site=site, method="bilinear", out_format=ofmt,
                                  ow=ow, oh=oh, in_sha256=hashlib.sha256(raw).hexdigest(), out_sha256=hashlib.sha256(out).hexdigest()))
                t += 1
    arrays["manifest"] = np.frombuffer(json.dumps(cases).encode(), np.uint8)
    np.savez_compressed(os.path.join(GOLD, "convertscale_gst114_packedout.npz"), **arrays)
    print("wrote", len(cases), "packed-output cases")


def gen_ties():
    """sizes whose 2-tap weights fall on exact .5 ties of the 8-bit (vertical) or 6-bit (packed / NV12-chroma horizontal)
    quantiser, where GstVideoScaler's bisection - not round-half-up - decides: tests/golden/convertscale_gst114_ties.npz"""
    cases, arrays = [], {}
    rng = np.random.default_rng(20261008)
    todo = [("BGRA", "BGRA", 8, 67, 8, 256), ("RGBA", "BGRA", 8, 13, 8, 768), ("BGRA", "RGBA", 45, 67, 20, 64), ("NV12", "BGRA", 16, 67, 16, 256),
            ("I420", "RGBA", 12, 21, 12, 448), ("NV12", "NV12", 134, 20, 128, 20), ("NV12", "NV12", 134, 134, 128, 512), ("I420", "I420", 90, 134, 128, 512),
            ("BGRA", "NV12", 134, 67, 128, 256), ("NV12", "I420", 42, 26, 448, 64), ("YUY2", "YUY2", 67, 10, 64, 10), ("UYVY", "UYVY", 67, 67, 64, 256),
            ("BGRA", "UYVY", 45, 8, 64, 8), ("I420", "YUY2", 21, 13, 448, 24), ("UYVY", "NV12", 134, 67, 128, 256), ("YUY2", "BGRA", 16, 67, 16, 256)]
    with tempfile.TemporaryDirectory() as tmp:
        exe = build_helper(tmp)
        for t, (ifmt, ofmt, w, h, ow, oh) in enumerate(todo):
            col, site = ["bt601", "bt709"][t % 2], ["jpeg", "mpeg2"][(t // 2) % 2]
            size = {"BGRA": w * h * 4, "RGBA": w * h * 4, "NV12": nv12_layout(w, h)[3], "I420": i420_layout(w, h)[4]}.get(ifmt, r4(2 * w) * h)
            raw = rng.integers(0, 256, size, dtype=np.uint8).tobytes()
            incaps = f"video/x-raw,format={ifmt},width={w},height={h},framerate=1/1"
            if ifmt not in ("BGRA", "RGBA"):
                incaps += f",colorimetry={col},chroma-site={site}"
            outcaps = f"video/x-raw,format={ofmt},width={ow},height={oh}"
            if ofmt not in ("BGRA", "RGBA"):
                outcaps += f",colorimetry={col},chroma-site={site}"
            out = gst_run(exe, tmp, raw, len(raw), incaps, "videoconvert ! videoscale", outcaps)
            name = f"tie_{ifmt.lower()}_to_{ofmt.lower()}_{t:02d}_{w}x{h}_to_{ow}x{oh}"
            arrays[name + "_in"], arrays[name + "_out"] = np.frombuffer(raw, np.uint8), np.frombuffer(out, np.uint8)
            cases.append(dict(name=name, in_format=ifmt, w=w, h=h, colorimetry=col, chroma_site=site, method="bilinear", out_format=ofmt,
                              ow=ow, oh=oh, in_sha256=hashlib.sha256(raw).hexdigest(), out_sha256=hashlib.sha256(out).hexdigest()))
    arrays["manifest"] = np.frombuffer(json.dumps(cases).encode(), np.uint8)
    np.savez_compressed(os.path.join(GOLD, "convertscale_gst114_ties.npz"), **arrays)
    print("wrote", len(cases), "tie cases")


def gen_yuv_nearest():
    """`videoscale method=nearest-neighbour` with YUV outputs (4:2:0 and packed 4:2:2) from all six input formats ->
    tests/golden/convertscale_gst114_yuvnearest.npz"""
    cases, arrays = [], {}
    rng = np.random.default_rng(20261009)
    cols, sites = ["bt601", "bt709", "bt2020"], ["jpeg", "mpeg2"]
    fixed = [(64, 36, 32, 18), (64, 36, 100, 50), (33, 17, 16, 9), (17, 9, 40, 21), (48, 40, 20, 37), (31, 30, 31, 12), (50, 20, 125, 20), (35, 29, 18, 8)]
    with tempfile.TemporaryDirectory() as tmp:
        exe = build_helper(tmp)
        t = 0
        for ifmt in ["BGRA", "RGBA", "NV12", "I420", "UYVY", "YUY2"]:
            for ofmt in ["NV12", "I420", "UYVY", "YUY2"]:
                for (w, h, ow, oh) in [fixed[t % 8], tuple(int(v) for v in rng.integers(8, 90, 4))]:
                    col, site = cols[t % 3], sites[(t // 3) % 2]
                    size = {"BGRA": w * h * 4, "RGBA": w * h * 4, "NV12": nv12_layout(w, h)[3], "I420": i420_layout(w, h)[4]}.get(ifmt, r4(2 * w) * h)
                    raw = rng.integers(0, 256, size, dtype=np.uint8).tobytes()
                    incaps = f"video/x-raw,format={ifmt},width={w},height={h},framerate=1/1"
                    if ifmt not in ("BGRA", "RGBA"):
                        incaps += f",colorimetry={col},chroma-site={site}"
                    out = gst_run(exe, tmp, raw, len(raw), incaps, "videoconvert ! videoscale method=nearest-neighbour",
                                  f"video/x-raw,format={ofmt},width={ow},height={oh},colorimetry={col},chroma-site={site}")
                    name = f"nn_{ifmt.lower()}_to_{ofmt.lower()}_{t:03d}_{w}x{h}_to_{ow}x{oh}"
                    arrays[name + "_in"], arrays[name + "_out"] = np.frombuffer(raw, np.uint8), np.frombuffer(out, np.uint8)
                    cases.append(dict(name=name, in_format=ifmt, w=w, h=h, colorimetry=col, chroma_site=site, method="nearest", out_format=ofmt,
                                      ow=ow, oh=oh, in_sha256=hashlib.sha256(raw).hexdigest(), out_sha256=hashlib.sha256(out).hexdigest()))
                    t += 1
    arrays["manifest"] = np.frombuffer(json.dumps(cases).encode(), np.uint8)
    np.savez_compressed(os.path.join(GOLD, "convertscale_gst114_yuvnearest.npz"), **arrays)
    print("wrote", len(cases), "nearest YUV-output cases")


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "yuvnearest":
    gen_yuv_nearest()
    sys.exit(0)
def gen_yuv_cubic():
    """`videoscale method=catrom` with YUV outputs (4:2:0 and packed 4:2:2) -> tests/golden/convertscale_gst114_yuvcubic.npz"""
    cases, arrays = [], {}
    rng = np.random.default_rng(20261010)
    cols, sites = ["bt601", "bt709", "bt2020"], ["jpeg", "mpeg2"]
    fixed = [(64, 36, 32, 18), (64, 36, 100, 50), (66, 34, 33, 17), (34, 18, 80, 42), (48, 40, 20, 38), (62, 60, 62, 24), (50, 20, 125, 20), (70, 58, 36, 16),
             (33, 17, 16, 9), (35, 29, 18, 8), (31, 30, 13, 12), (128, 72, 40, 24)]
    with tempfile.TemporaryDirectory() as tmp:
        exe = build_helper(tmp)
        t = 0
        for ifmt in ["BGRA", "NV12", "I420", "UYVY", "YUY2", "RGBA"]:
            for ofmt in ["NV12", "I420", "UYVY", "YUY2"]:
                for (w, h, ow, oh) in [fixed[t % 12], tuple(int(v) for v in rng.integers(24, 100, 4))]:
                    col, site = cols[t % 3], sites[(t // 3) % 2]
                    size = {"BGRA": w * h * 4, "RGBA": w * h * 4, "NV12": nv12_layout(w, h)[3], "I420": i420_layout(w, h)[4]}.get(ifmt, r4(2 * w) * h)
                    raw = rng.integers(0, 256, size, dtype=np.uint8).tobytes()
                    incaps = f"video/x-raw,format={ifmt},width={w},height={h},framerate=1/1"
                    if ifmt not in ("BGRA", "RGBA"):
                        incaps += f",colorimetry={col},chroma-site={site}"
                    out = gst_run(exe, tmp, raw, len(raw), incaps, "videoconvert ! videoscale method=catrom",
                                  f"video/x-raw,format={ofmt},width={ow},height={oh},colorimetry={col},chroma-site={site}")
                    name = f"cub_{ifmt.lower()}_to_{ofmt.lower()}_{t:03d}_{w}x{h}_to_{ow}x{oh}"
                    arrays[name + "_in"], arrays[name + "_out"] = np.frombuffer(raw, np.uint8), np.frombuffer(out, np.uint8)
                    cases.append(dict(name=name, in_format=ifmt, w=w, h=h, colorimetry=col, chroma_site=site, method="bicubic", out_format=ofmt,
                                      ow=ow, oh=oh, in_sha256=hashlib.sha256(raw).hexdigest(), out_sha256=hashlib.sha256(out).hexdigest()))
                    t += 1
    arrays["manifest"] = np.frombuffer(json.dumps(cases).encode(), np.uint8)
    np.savez_compressed(os.path.join(GOLD, "convertscale_gst114_yuvcubic.npz"), **arrays)
    print("wrote", len(cases), "bicubic YUV-output cases")


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "yuvcubic":
    gen_yuv_cubic()
    sys.exit(0)
def gen_mixed_sites():
    """YUV -> YUV with DIFFERENT chroma sitings on the two sides, where that is pinned (NV12 <-> packed resample with both; the
    I420 <-> packed fast paths and the UYVY <-> YUY2 swizzle ignore it): tests/golden/convertscale_gst114_mixedsite.npz"""
    cases, arrays = [], {}
    rng = np.random.default_rng(20261011)
    todo = [("NV12", "UYVY", 64, 36, 64, 36), ("NV12", "YUY2", 66, 34, 33, 17), ("UYVY", "NV12", 64, 36, 64, 36), ("YUY2", "NV12", 50, 20, 125, 20),
            ("I420", "UYVY", 48, 40, 20, 38), ("YUY2", "I420", 33, 17, 33, 17), ("UYVY", "YUY2", 62, 30, 31, 12), ("NV12", "UYVY", 35, 29, 80, 41),
            ("NV12", "NV12", 64, 36, 40, 30), ("I420", "I420", 33, 17, 33, 17), ("UYVY", "UYVY", 64, 36, 100, 50), ("YUY2", "YUY2", 34, 18, 34, 18)]
    with tempfile.TemporaryDirectory() as tmp:
        exe = build_helper(tmp)
        for t, (ifmt, ofmt, w, h, ow, oh) in enumerate(todo):
            col = ["bt601", "bt709", "bt2020"][t % 3]
            si, so = (("jpeg", "mpeg2"), ("mpeg2", "jpeg"))[t % 2]
            size = {"NV12": nv12_layout(w, h)[3], "I420": i420_layout(w, h)[4]}.get(ifmt, r4(2 * w) * h)
            raw = rng.integers(0, 256, size, dtype=np.uint8).tobytes()
            out = gst_run(exe, tmp, raw, len(raw), f"video/x-raw,format={ifmt},width={w},height={h},framerate=1/1,colorimetry={col},chroma-site={si}",
                          "videoconvert ! videoscale", f"video/x-raw,format={ofmt},width={ow},height={oh},colorimetry={col},chroma-site={so}")
            name = f"ms_{ifmt.lower()}_to_{ofmt.lower()}_{t:02d}_{w}x{h}_to_{ow}x{oh}_{si}_{so}"
            arrays[name + "_in"], arrays[name + "_out"] = np.frombuffer(raw, np.uint8), np.frombuffer(out, np.uint8)
            cases.append(dict(name=name, in_format=ifmt, w=w, h=h, colorimetry=col, chroma_site=si, out_chroma_site=so, method="bilinear", out_format=ofmt,
                              ow=ow, oh=oh, in_sha256=hashlib.sha256(raw).hexdigest(), out_sha256=hashlib.sha256(out).hexdigest()))
    arrays["manifest"] = np.frombuffer(json.dumps(cases).encode(), np.uint8)
    np.savez_compressed(os.path.join(GOLD, "convertscale_gst114_mixedsite.npz"), **arrays)
    print("wrote", len(cases), "mixed-siting cases")


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "mixedsite":
    gen_mixed_sites()
    sys.exit(0)
def gen_remat():
    """YUV -> YUV with a MATRIX change (all four YUV formats either side) and NV12 <-> I420 with a SITING change, same size and
    scaled, through the real `videoconvert ! videoscale` -> tests/golden/convertscale_gst114_remat.npz"""
    cases, arrays = [], {}
    rng = np.random.default_rng(20261012)
    cols, sites, fm = ["bt601", "bt709", "bt2020"], ["jpeg", "mpeg2"], ["NV12", "I420", "UYVY", "YUY2"]
    sizes = [(64, 36, 64, 36), (33, 17, 33, 17), (34, 19, 34, 19), (64, 36, 40, 30), (66, 34, 33, 17), (35, 29, 80, 41), (48, 40, 48, 40), (31, 30, 31, 30),
             (128, 72, 64, 36), (20, 13, 20, 13)]
    with tempfile.TemporaryDirectory() as tmp:
        exe = build_helper(tmp)
        t = 0
        for ifmt in fm:
            for ofmt in fm:
                for rep in range(4):
                    w, h, ow, oh = sizes[(t * 3 + rep) % len(sizes)]
                    ci = cols[t % 3]
                    co = cols[(t + 1 + rep % 2) % 3] if rep < 3 else ci            # the 4th case of a pair: siting change only
                    si, so = sites[(t + rep) % 2], sites[(t + rep + (1 if rep in (1, 3) else 0)) % 2]
                    t += 1
                    in420, out420 = ifmt in ("NV12", "I420"), ofmt in ("NV12", "I420")
                    if ci == co and not (in420 and out420 and ifmt != ofmt and si != so):
                        continue                                                  # covered by the other fixture sets (or passthrough)
                    size = {"NV12": nv12_layout(w, h)[3], "I420": i420_layout(w, h)[4]}.get(ifmt, r4(2 * w) * h)
                    raw = rng.integers(0, 256, size, dtype=np.uint8).tobytes()
                    out = gst_run(exe, tmp, raw, len(raw), f"video/x-raw,format={ifmt},width={w},height={h},framerate=1/1,colorimetry={ci},chroma-site={si}",
                                  "videoconvert ! videoscale", f"video/x-raw,format={ofmt},width={ow},height={oh},colorimetry={co},chroma-site={so}")
                    name = f"rm_{ifmt.lower()}_to_{ofmt.lower()}_{len(cases):02d}_{w}x{h}_to_{ow}x{oh}_{ci}_{si}_{co}_{so}"
                    arrays[name + "_in"], arrays[name + "_out"] = np.frombuffer(raw, np.uint8), np.frombuffer(out, np.uint8)
                    cases.append(dict(name=name, in_format=ifmt, w=w, h=h, colorimetry=ci, chroma_site=si, out_colorimetry=co, out_chroma_site=so, method="bilinear",
                                      out_format=ofmt, ow=ow, oh=oh, in_sha256=hashlib.sha256(raw).hexdigest(), out_sha256=hashlib.sha256(out).hexdigest()))
    arrays["manifest"] = np.frombuffer(json.dumps(cases).encode(), np.uint8)
    np.savez_compressed(os.path.join(GOLD, "convertscale_gst114_remat.npz"), **arrays)
    print("wrote", len(cases), "matrix / siting change cases")


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "remat":
    gen_remat()
    sys.exit(0)
if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "ties":
    gen_ties()
    sys.exit(0)
if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "packedout":
    gen_packed_outputs()
    sys.exit(0)
if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "packed":
    gen_packed()
    sys.exit(0)
if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "bicubic":
    gen_bicubic()
    sys.exit(0)
if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "yuvout":
        gen_yuv_outputs()
    else:
        main()
