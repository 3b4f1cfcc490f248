#!/usr/bin/env python3
"""One-off fuzz of the gst-exact convertscale cells on the GPU against the oracle: random formats, sizes, methods, matrices,
sitings.  Prints every mismatch; exit status 1 if there is one.  usage: fuzz_gst_exact.py [cases] [seed]"""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gstreamer-metal_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import oracle_lib  # noqa: E402
import vfhip  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 400
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
orc = oracle_lib.load()
bad = 0
kernels = {}
for case in range(N):
    ifmt = ["NV12", "I420", "BGRA", "RGBA", "UYVY", "YUY2"][rng.integers(6)]
    ofmt = ["BGRA", "RGBA", "NV12", "I420", "UYVY", "YUY2"][rng.integers(6)]
    method = ["bilinear", "nearest", "bicubic"][rng.integers(3)]
    big = rng.integers(4) == 0
    w, h, ow, oh = (int(v) for v in rng.integers(2, 700 if big else 120, 4))
    if rng.integers(5) == 0:
        ow, oh = max(w // 2, 1), max(h // 2, 1)                     # exact halves: the fast paths
        w, h = 2 * ow, 2 * oh
    if rng.integers(8) == 0:
        w = max(16, w & ~7); ow, oh = w, h                          # conversion at the same size, width % 8 == 0: the k_cs_*_same kernels (YUV in, RGB out)
    if ofmt in ("NV12", "I420"):
        h, oh = max(h, 8), max(oh, 8)                                # GStreamer 1.14 mishandles tiny 4:2:0 outputs
    if method == "bicubic" and not all(i == o or math.ceil(4 * max(1.0, i / o)) <= min(i, 64) for i, o in ((w, ow), (h, oh))):
        method = "bilinear"
    col, site = ["bt601", "bt709", "bt2020"][rng.integers(3)], ["jpeg", "mpeg2"][rng.integers(2)]
    raw = rng.integers(0, 256, vfhip.plane_layout(ifmt, w, h)[1], dtype=np.uint8)
    borders = rng.integers(5) == 0 and not (w * oh == h * ow)
    cs = vfhip.ConvertScale(0)
    try:
        cs.configure(ifmt, w, h, ofmt, ow, oh, method=method, colorimetry=col, chroma_site=site, add_borders=bool(borders), border_color=0xC0123456)
        got = cs.process(raw)
        k = cs.kernel_name
    except vfhip.VfHipError as e:
        # bicubic on a line shorter than its filter (here: a chroma line): refused — the oracle must refuse the same case
        assert e.code == -2 and method == "bicubic", (e, ifmt, (w, h), ofmt, (ow, oh), method)
        try:
            if borders:
                if oracle_lib.convertscale_with_borders(orc, ifmt, w, h, raw, col, site, method, ofmt, ow, oh, 0xC0123456) is None:
                    raise RuntimeError("unaligned rectangle: bicubic has no metal fallback")
            else:
                orc.convertscale(ifmt, w, h, raw, col, site, method, ofmt, ow, oh)
            bad += 1
            print("REFUSED BY THE LIBRARY ONLY", ifmt, (w, h), "->", ofmt, (ow, oh), flush=True)
        except RuntimeError:
            kernels["refused"] = kernels.get("refused", 0) + 1
        continue
    finally:
        cs.close()
    kernels[k] = kernels.get(k, 0) + 1
    if k == "k_cs_metal":
        continue                                                     # an unpinned cell (not gst-exact): nothing to compare
    try:
        want = (oracle_lib.convertscale_with_borders(orc, ifmt, w, h, raw, col, site, method, ofmt, ow, oh, 0xC0123456) if borders
                else orc.convertscale(ifmt, w, h, raw, col, site, method, ofmt, ow, oh))
    except RuntimeError:
        if borders:
            kernels["borders: rectangle outside the bicubic domain"] = kernels.get("borders: rectangle outside the bicubic domain", 0) + 1
            bad += 1
            print("LIBRARY ACCEPTED WHAT THE ORACLE REFUSES", ifmt, (w, h), "->", ofmt, (ow, oh), method, flush=True)
            continue
        raise
    if want is None:
        bad += 1
        print("EXACT KERNEL ON AN UNALIGNED BORDER RECTANGLE", ifmt, (w, h), "->", ofmt, (ow, oh), k, flush=True)
        continue
    if borders:
        kernels["(with borders)"] = kernels.get("(with borders)", 0) + 1
    if not np.array_equal(np.asarray(got).reshape(-1), np.asarray(want).reshape(-1)):
        bad += 1
        d = (np.asarray(got).reshape(-1) != np.asarray(want).reshape(-1)).sum()
        # which side is it?  run both again: a deterministic difference shows again on both, a one-off points at whichever side changed
        cs2 = vfhip.ConvertScale(0)
        cs2.configure(ifmt, w, h, ofmt, ow, oh, method=method, colorimetry=col, chroma_site=site, add_borders=bool(borders), border_color=0xC0123456)
        got2 = cs2.process(raw); cs2.close()
        want2 = (oracle_lib.convertscale_with_borders(orc, ifmt, w, h, raw, col, site, method, ofmt, ow, oh, 0xC0123456) if borders
                 else orc.convertscale(ifmt, w, h, raw, col, site, method, ofmt, ow, oh))
        # the whole record goes to a file: a one-off must be diagnosable from what this run saw, without trying to make it happen again
        dump_dir = os.path.join(ROOT, "gpurun_out", "fuzz_dumps")
        os.makedirs(dump_dir, exist_ok=True)
        dump = os.path.join(dump_dir, f"gst_seed{sys.argv[2] if len(sys.argv) > 2 else 1}_case{case}.npz")
        np.savez_compressed(dump, raw=raw, got=np.asarray(got).reshape(-1), want=np.asarray(want).reshape(-1), got2=np.asarray(got2).reshape(-1),
                            want2=np.asarray(want2).reshape(-1),
                            params=np.array([ifmt, str(w), str(h), ofmt, str(ow), str(oh), method, col, site, k, str(bool(borders)),
                                             os.environ.get("VFHIP_DEBUG_POISON", ""), os.environ.get("VFHIP_ORACLE_THREADS", "1")]))
        print("MISMATCH", ifmt, (w, h), "->", ofmt, (ow, oh), method, col, site, k, d, "bytes", "borders" if borders else "", "| record:", dump,
              "| library repeats itself:", bool(np.array_equal(np.asarray(got).reshape(-1), np.asarray(got2).reshape(-1))),
              "| oracle repeats itself:", bool(np.array_equal(np.asarray(want).reshape(-1), np.asarray(want2).reshape(-1))),
              "| second run equal:", bool(np.array_equal(np.asarray(got2).reshape(-1), np.asarray(want2).reshape(-1))), flush=True)
print("cases", N, "mismatches", bad, "kernels", kernels)
sys.exit(1 if bad else 0)
