import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "gstreamer-metal_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, oracle_lib, vfhip
orc = oracle_lib.load()
rng = np.random.default_rng(5)
for (ifmt, w, h, ofmt, ow, oh, col, site) in [("I420", 54, 67, "RGBA", 18, 74, "bt601", "jpeg"), ("I420", 93, 32, "RGBA", 33, 101, "bt709", "mpeg2"),
                                              ("NV12", 54, 67, "RGBA", 18, 74, "bt601", "jpeg"), ("BGRA", 54, 67, "RGBA", 18, 74, "bt601", "jpeg"),
                                              ("I420", 54, 67, "RGBA", 18, 67, "bt601", "jpeg"), ("I420", 54, 67, "RGBA", 54, 74, "bt601", "jpeg")]:
    raw = rng.integers(0, 256, vfhip.plane_layout(ifmt, w, h)[1], dtype=np.uint8)
    cs = vfhip.ConvertScale(0)
    cs.configure(ifmt, w, h, ofmt, ow, oh, method="bicubic", colorimetry=col, chroma_site=site)
    got = cs.process(raw).reshape(oh, ow, 4)
    want = orc.convertscale(ifmt, w, h, raw, col, site, "bicubic", ofmt, ow, oh).reshape(oh, ow, 4)
    d = got.astype(int) - want.astype(int)
    ys, xs, cs_ = np.nonzero(d)
    print(ifmt, (w, h), "->", (ow, oh), cs.kernel_name, "diff bytes", len(ys), "channels", np.unique(cs_), "rows", np.unique(ys)[:10], "cols", np.unique(xs)[:10],
          "sample", [(int(got[y, x, c]), int(want[y, x, c])) for y, x, c in list(zip(ys, xs, cs_))[:6]], flush=True)
    cs.close()
