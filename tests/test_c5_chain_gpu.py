"""BASELINE configs[4] (C5) at its real size: `vfhipdeinterlace method=greedyh` on NV12 3840x2160 feeding
`vfhipconvertscale` -> BGRA 1920x1080, one such stream per GPU (SURVEY.md §8d "C5 per stream-frame").
Reference for the work replaced: deinterlace/metaldeinterlace_shaders.h:181-218 (greedyH), :88-148 (bob / linear),
:150-179 (weave); deinterlace/metaldeinterlacerenderer.m:295-413 (passes + history); convertscale/metalconvertscale_shaders.h:91-116.

Oracles: the deinterlacer against oracle/metalref.c (+-1 LSB, PARITY UNPINNED vs real Metal — see that file's header),
the convert+scale leg against oracle/gst114.c (bit-exact, pinned on real GStreamer 1.14 vectors).  All through the C ABI,
device-resident frames, the batched entry points the bench uses."""
import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.gpu
W, H, OW, OH = 3840, 2160, 1920, 1080
TOL = 1
CHAIN_TOL = 4


def interlaced_stream(n, seed=0):
    """n NV12 2160p frames of one stream: smooth static background (greedyh weaves it) + a band that moves 24 pixels per
    frame (greedyh must bob there) + per-frame noise below / above the motion threshold on the two halves."""
    rng = np.random.default_rng(seed)
    pl, size = ol.raw_layout("NV12", W, H)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    base_y = 128 + 80 * np.sin(xx / 37.0) * np.cos(yy / 23.0)
    cy, cx = np.mgrid[0:H // 2, 0:W].astype(np.float32)
    base_c = 128 + 60 * np.sin(cx / 51.0 + 1.0) * np.cos(cy / 19.0)
    frames = []
    for k in range(n):
        y = base_y.copy()
        y[:, W // 2:] = np.roll(base_y[:, W // 2:], 24 * k, axis=1)                 # motion on the right half
        y[H // 3:H // 3 + 200, :] = 40 + 170 * ((xx[H // 3:H // 3 + 200, :] + 24 * k) % 97 > 48)   # hard moving edges
        y += rng.integers(-3, 4, y.shape)
        c = base_c + rng.integers(-2, 3, base_c.shape)
        raw = np.empty(size, np.uint8)
        raw[pl[0][0]:pl[0][0] + W * H] = np.clip(y, 0, 255).astype(np.uint8).reshape(-1)
        raw[pl[1][0]:pl[1][0] + W * H // 2] = np.clip(c, 0, 255).astype(np.uint8).reshape(-1)
        frames.append(raw)
    return frames


def ring(frames, pitch):
    import torch
    buf = np.zeros((len(frames), pitch), np.uint8)
    for k, f in enumerate(frames):
        buf[k, :f.size] = f
    return torch.from_numpy(buf).cuda()


def close(got, want, what, max_off_by_one=0.02):
    d = np.abs(got.astype(np.int16) - want.astype(np.int16))
    assert d.max() <= TOL, f"{what}: max diff {d.max()} at byte {int(np.argmax(d))} ({int((d > TOL).sum())} bytes beyond +-{TOL})"
    assert (d > 0).mean() <= max_off_by_one, f"{what}: {(d > 0).mean():.4f} of bytes differ by 1"


@pytest.fixture(scope="module")
def stream4():
    return interlaced_stream(4, seed=5)


@pytest.mark.parametrize("method,tff", [("greedyh", True), ("greedyh", False), ("bob", True), ("weave", False), ("linear", True)])
def test_deinterlace_2160p_batch_vs_oracle(vfhip, metalref, stream4, method, tff):
    """full 2160p frames, 4 consecutive frames of one stream in ONE batched launch (history = previous frame of the
    batch): every byte of every output frame against the oracle, and batch == the same frames pushed one at a time"""
    import torch
    n = len(stream4)
    size = stream4[0].size
    pitch = (size + 255) // 256 * 256
    din = ring(stream4, pitch)
    dout = torch.zeros((n, pitch), dtype=torch.uint8, device="cuda")
    done = torch.zeros((n, pitch), dtype=torch.uint8, device="cuda")
    s = torch.cuda.Stream()
    torch.cuda.synchronize()
    d = vfhip.Deinterlace(0)
    d.configure("NV12", W, H)
    d.process_device(din.data_ptr(), dout.data_ptr(), method=method, tff=tff, threshold=0.1, stream=s.cuda_stream, n_frames=n, in_pitch=pitch, out_pitch=pitch)
    d1 = vfhip.Deinterlace(0)
    d1.configure("NV12", W, H)
    for k in range(n):
        d1.process_device(din[k].data_ptr(), done[k].data_ptr(), method=method, tff=tff, threshold=0.1, stream=s.cuda_stream)
    s.synchronize()
    assert torch.equal(dout, done), "a batch of n consecutive frames must equal n single-frame calls on one handle"
    out = dout.cpu().numpy()
    code = vfhip.DEINTERLACE_METHODS[method]
    moved = 0.0
    for k in range(n):
        want = metalref.deinterlace("NV12", W, H, stream4[k], stream4[k - 1] if k else None, code, tff=tff, threshold=0.1)
        close(out[k, :size], want, f"2160p {method} tff={tff} frame {k}")
        if k and method == "greedyh":
            weave = metalref.deinterlace("NV12", W, H, stream4[k], stream4[k - 1], vfhip.DEINTERLACE_METHODS["weave"], tff=tff, threshold=0.1)
            moved = max(moved, float((want[:W * H] != weave[:W * H]).mean()))
    if method == "greedyh":
        assert 0.02 < moved < 0.6, f"the test stream must exercise both greedyh branches (bob fraction {moved:.3f})"
    d.close()
    d1.close()


def test_c5_chain_2160p_device_resident(vfhip, metalref, oracle, stream4):
    """configs[4], one stream: deinterlace greedyh (NV12 2160p) -> convertscale (BGRA 1080p), the intermediate frames stay
    in HBM, both legs batched over 4 consecutive frames on one HIP stream.
      leg 1 vs oracle/metalref.c: +-1 LSB;  leg 2 vs oracle/gst114.c on the SAME intermediate bytes: bit-exact;
      whole chain vs oracle(oracle(x)): leg 1's +-1 LSB in a Y byte (gain 298/256) and in a U / V byte (gain up to
      548/256 through the bt2020 matrix) can add up to 4 LSB in an RGB channel before the averaging taps, so CHAIN_TOL = 4;
      the fraction of differing bytes stays tiny because leg 1 is byte-identical almost everywhere."""
    import torch
    n = len(stream4)
    size = stream4[0].size
    pitch = (size + 255) // 256 * 256
    opitch = OW * OH * 4
    din = ring(stream4, pitch)
    dmid = torch.zeros((n, pitch), dtype=torch.uint8, device="cuda")
    dout = torch.zeros((n, opitch), dtype=torch.uint8, device="cuda")
    s = torch.cuda.Stream()
    torch.cuda.synchronize()
    d = vfhip.Deinterlace(0)
    d.configure("NV12", W, H)
    cs = vfhip.ConvertScale(0)
    cs.configure("NV12", W, H, "BGRA", OW, OH, method="bilinear", numerics="gst-exact", colorimetry="bt2020", chroma_site="mpeg2")
    assert cs.kernel_name == "k_cs_nv12_half"
    d.process_device(din.data_ptr(), dmid.data_ptr(), method="greedyh", tff=True, threshold=0.1, stream=s.cuda_stream, n_frames=n, in_pitch=pitch, out_pitch=pitch)
    cs.process_device(dmid.data_ptr(), dout.data_ptr(), stream=s.cuda_stream, n_frames=n, in_pitch=pitch, out_pitch=opitch)
    s.synchronize()
    mid, out = dmid.cpu().numpy(), dout.cpu().numpy()
    for k in range(n):
        want_mid = metalref.deinterlace("NV12", W, H, stream4[k], stream4[k - 1] if k else None, 3, tff=True, threshold=0.1)
        close(mid[k, :size], want_mid, f"C5 leg 1 frame {k}")
        exact = oracle.convertscale("NV12", W, H, mid[k, :size], "bt2020", "mpeg2", "bilinear", "BGRA", OW, OH)
        assert np.array_equal(out[k].reshape(OH, OW, 4), exact), f"C5 leg 2 frame {k}: convertscale of the device intermediate is not bit-exact"
        chain = oracle.convertscale("NV12", W, H, want_mid, "bt2020", "mpeg2", "bilinear", "BGRA", OW, OH)
        dd = np.abs(out[k].reshape(OH, OW, 4).astype(np.int16) - chain.astype(np.int16))
        assert dd.max() <= CHAIN_TOL, f"C5 chain frame {k}: max diff {dd.max()} vs oracle(oracle(x))"
        assert (dd > 0).mean() < 0.02
    d.close()
    cs.close()


def test_c5_chain_through_host_api(vfhip, metalref, oracle, stream4):
    """the same chain through the synchronous entry points the element shells call (`*_process`, host frames)"""
    d = vfhip.Deinterlace(0)
    d.configure("NV12", W, H)
    cs = vfhip.ConvertScale(0)
    cs.configure("NV12", W, H, "BGRA", OW, OH, colorimetry="bt2020", chroma_site="mpeg2")
    prev = None
    for k, f in enumerate(stream4[:2]):
        mid = d.process(f, method="greedyh", tff=True, threshold=0.1)
        close(mid, metalref.deinterlace("NV12", W, H, f, prev, 3, tff=True, threshold=0.1), f"host C5 leg 1 frame {k}")
        out = cs.process(mid).reshape(OH, OW, 4)
        assert np.array_equal(out, oracle.convertscale("NV12", W, H, mid, "bt2020", "mpeg2", "bilinear", "BGRA", OW, OH))
        prev = f
    d.close()
    cs.close()


def test_deinterlace_strip_heights_identical():
    """k_deinterlace_420q walks 8-row strips for a frame and 16 / 32-row strips for big batches (deint_launch): the strip height is a
    speed matter only.  $VFHIP_DEINT_ROWS is read once per process, so each height runs tools/exp/deint_hash.py --quick (NV12 / I420,
    both field orders, bob / weave / greedy-H, 1080p and sizes that leave partial strips) in a process of its own; the digests over
    all output frames must be equal."""
    import os
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    digests = {}
    for rows in ("", "2", "16", "32"):
        env = dict(os.environ)
        env.pop("VFHIP_DEINT_ROWS", None)
        if rows:
            env["VFHIP_DEINT_ROWS"] = rows
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "exp", "deint_hash.py"), "--quick"], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        m = re.search(r"frames (\d+) sha256 ([0-9a-f]{64})", r.stdout)
        assert m and int(m.group(1)) > 100, r.stdout[-500:]
        digests[rows or "default (8)"] = m.group(2)
    assert len(set(digests.values())) == 1, digests
