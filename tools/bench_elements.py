#!/usr/bin/env python3
"""tools/bench_elements.py — kernel-only throughput of the other BASELINE configs (device-resident frames, HIP events
on the launch stream): C3 videofilter BGRA 1080p all 15 properties, C4 compositor 4xBGRA 1080p + NV12 720p -> 2160p,
C5 greedy-H deinterlace NV12 2160p (+ the convertscale of the chain), plus the generic convertscale kernel on C1.
Not the headline bench (that is bench.py); one JSON line per config."""
import json
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gstreamer-metal_amd"))
import torch  # noqa: E402
import vfhip  # noqa: E402

PEAK = 8000.0


def timed(fn, stream, iters, warm=3):
    for _ in range(warm):
        fn()
    stream.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(iters):
        fn()
    e1.record(stream)
    stream.synchronize()
    return e0.elapsed_time(e1) / iters


def report(name, kernel, ms, frames, bytes_per_frame):
    gbs = bytes_per_frame * frames / (ms * 1e-3) / 1e9
    print(json.dumps({"config": name, "kernel": kernel, "ms_per_launch": round(ms, 4), "frames_per_launch": frames,
                      "frames_per_s": round(frames / ms * 1e3, 1), "algorithmic_bytes_per_frame": bytes_per_frame,
                      "achieved_GBps": round(gbs, 1), "frac_of_8TBps": round(gbs / PEAK, 4)}), flush=True)


def ring(n, size, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return torch.randint(0, 256, (n, (size + 255) // 256 * 256), dtype=torch.uint8, device="cuda", generator=g)


def main():
    s = torch.cuda.Stream()
    R = 24   # distinct frames per config (>= 400 MB working set with outputs; kernels here are far from the roofline anyway)

    # C3: videofilter BGRA 1080p, all 15 properties (SURVEY.md §8d parameter set)
    w, h = 1920, 1080
    fin, fout = ring(R, 4 * w * h, 1), ring(R, 4 * w * h, 2)
    vf = vfhip.VideoFilter(0)
    vf.configure("BGRA", w, h)
    n = 33
    import numpy as np
    g = np.linspace(0, 1, n, dtype=np.float32)
    lut = np.ones((n, n, n, 4), np.float32)
    lut[..., 0], lut[..., 1], lut[..., 2] = g[None, None, :] ** 1.05, g[None, :, None], g[:, None, None] ** 0.95
    vf.set_lut(lut)
    prm = vfhip.filter_params(brightness=0.1, contrast=1.2, saturation=0.8, hue=0.3 * math.pi, gamma=1.5, sharpness=0.5, sepia=0.2,
                              noise=0.1, vignette=0.3, invert=True, chroma_key=(0.0, 1.0, 0.0), tolerance=0.3, smoothness=0.1)

    vp = fin.shape[1]

    def c3():
        vf.process_device(fin.data_ptr(), fout.data_ptr(), prm, stream=s.cuda_stream, n_frames=R, in_pitch=vp, out_pitch=vp)
    report("C3 videofilter BGRA 1080p all-15 + 33^3 LUT", "k_vf_sharp", timed(c3, s, 5), R, 2 * 4 * w * h)
    # the same on natural-like content (smooth gradients + a little noise): neighbouring pixels hit neighbouring LUT cells,
    # so the 8 gathers per pixel mostly share cache lines — uniform random bytes are the LUT stage's worst case
    yy, xx = torch.meshgrid(torch.arange(h, device="cuda"), torch.arange(w, device="cuda"), indexing="ij")
    smooth = torch.empty((R, h, w, 4), dtype=torch.uint8, device="cuda")
    for k in range(R):
        base = torch.stack([128 + 100 * torch.sin(xx / (90.0 + k)) * torch.cos(yy / 70.0), 128 + 90 * torch.cos(xx / 130.0 + k), 128 + 110 * torch.sin(yy / (50.0 + k)),
                            torch.full_like(xx, 255, dtype=torch.float32)], dim=-1)
        smooth[k] = (base + torch.randint(-3, 4, base.shape, device="cuda")).clamp(0, 255).to(torch.uint8)
    fsm = torch.zeros_like(fin)
    fsm[:, :4 * w * h] = smooth.reshape(R, -1)
    del smooth, yy, xx

    def c3s():
        vf.process_device(fsm.data_ptr(), fout.data_ptr(), prm, stream=s.cuda_stream, n_frames=R, in_pitch=vp, out_pitch=vp)
    report("C3 videofilter BGRA 1080p all-15 + 33^3 LUT, smooth (natural-like) input", "k_vf_sharp", timed(c3s, s, 5), R, 2 * 4 * w * h)
    del fsm
    prm0 = vfhip.filter_params(brightness=0.1, contrast=1.2, saturation=0.8, gamma=1.5)

    def c3b():
        vf.process_device(fin.data_ptr(), fout.data_ptr(), prm0, stream=s.cuda_stream, n_frames=R, in_pitch=vp, out_pitch=vp)
    vf.clear_lut()
    report("videofilter BGRA 1080p colour-only (no blur, no LUT)", "k_vf_point", timed(c3b, s, 5), R, 2 * 4 * w * h)
    vf.close()
    del fin, fout

    # C5a: greedy-H deinterlace NV12 2160p (cur + prev read, out written)
    w, h = 3840, 2160
    size = vfhip.plane_layout("NV12", w, h)[1]
    din, dout = ring(R, size, 3), ring(R, size, 4)
    d = vfhip.Deinterlace(0)
    d.configure("NV12", w, h)

    dp = din.shape[1]

    def c5():
        d.process_device(din.data_ptr(), dout.data_ptr(), method="greedyh", tff=True, threshold=0.1, stream=s.cuda_stream,
                         n_frames=R, in_pitch=dp, out_pitch=dp)
    report("C5a deinterlace greedyh NV12 2160p (batch = one stream's consecutive frames)", "k_deinterlace_420q", timed(c5, s, 5), R, 3 * size)
    d.close()
    del din, dout

    # C4: compositor 4 x BGRA 1080p (alpha .9, over) + NV12 720p centred (alpha .7) -> BGRA 2160p, black background
    ow, oh = 3840, 2160
    NC = 32
    quads = [ring(NC, 4 * 1920 * 1080, 10 + k) for k in range(4)]
    nv = ring(NC, vfhip.plane_layout("NV12", 1280, 720)[1], 20)
    out = ring(NC, 4 * ow * oh, 21)
    comp = vfhip.Compositor(0)
    comp.configure("BGRA", ow, oh)

    pads = [comp.pad("BGRA", 1920, 1080, quads[q].data_ptr(), (q % 2) * 1920, (q // 2) * 1080, 1920, 1080, 0.9, "over") for q in range(4)]
    pads.append(comp.pad("NV12", 1280, 720, nv.data_ptr(), (ow - 1280) // 2, (oh - 720) // 2, 1280, 720, 0.7, "over", "bt709"))
    pitches = [quads[q].shape[1] for q in range(4)] + [nv.shape[1]]

    def c4():
        comp.composite_device(pads, out.data_ptr(), background="black", stream=s.cuda_stream, n_frames=NC, pad_pitches=pitches, out_pitch=out.shape[1])
    report("C4 compositor 4xBGRA1080p + NV12 720p -> BGRA 2160p", "k_compositor_quads + k_compositor_420", timed(c4, s, 5), NC, 4 * 4 * 1920 * 1080 + 1280 * 720 * 3 // 2 + 4 * ow * oh)
    comp.close()
    del quads, nv, out

    # C1: generic gst-exact kernel, NV12 1080p -> BGRA 640x480
    w, h, ow, oh = 1920, 1080, 640, 480
    size = vfhip.plane_layout("NV12", w, h)[1]
    pitch = (size + 255) // 256 * 256
    cin, cout = ring(64, size, 30), ring(64, 4 * ow * oh, 31)
    cs = vfhip.ConvertScale(0)
    cs.configure("NV12", w, h, "BGRA", ow, oh, colorimetry="bt709", chroma_site="mpeg2")

    def c1():
        cs.process_device(cin.data_ptr(), cout.data_ptr(), stream=s.cuda_stream, n_frames=64, in_pitch=pitch, out_pitch=cout.shape[1])
    report("C1 convertscale NV12 1080p -> BGRA 640x480 (generic gst-exact)", cs.kernel_name, timed(c1, s, 5), 64, w * h * 3 // 2 + 4 * ow * oh)
    cs.close()

    # C2's shape with an I420 input: k_cs_i420_half (GStreamer's I420 path replicates chroma: less arithmetic per pixel)
    w, h, ow, oh = 3840, 2160, 1920, 1080
    size = vfhip.plane_layout("I420", w, h)[1]
    pitch = (size + 255) // 256 * 256
    NI = 64
    iin, iout = ring(NI, size, 50), ring(NI, 4 * ow * oh, 51)
    cs = vfhip.ConvertScale(0)
    cs.configure("I420", w, h, "BGRA", ow, oh, colorimetry="bt2020", chroma_site="mpeg2")

    def c2i():
        cs.process_device(iin.data_ptr(), iout.data_ptr(), stream=s.cuda_stream, n_frames=NI, in_pitch=pitch, out_pitch=iout.shape[1])
    report("C2 shape, I420 2160p -> BGRA 1080p bilinear (gst-exact)", cs.kernel_name, timed(c2i, s, 20, warm=10), NI, w * h * 3 // 2 + 4 * ow * oh)
    cs.close()
    del iin, iout

    # C2 with method=bicubic (videoscale method=catrom, bit-exact)
    w, h, ow, oh = 3840, 2160, 1920, 1080
    size = vfhip.plane_layout("NV12", w, h)[1]
    pitch = (size + 255) // 256 * 256
    NB = 16
    bin_, bout = ring(NB, size, 40), ring(NB, 4 * ow * oh, 41)
    cs = vfhip.ConvertScale(0)
    cs.configure("NV12", w, h, "BGRA", ow, oh, method="bicubic", colorimetry="bt2020", chroma_site="mpeg2")

    def c2b():
        cs.process_device(bin_.data_ptr(), bout.data_ptr(), stream=s.cuda_stream, n_frames=NB, in_pitch=pitch, out_pitch=bout.shape[1])
    report("C2 convertscale NV12 2160p -> BGRA 1080p method=bicubic (gst-exact catrom)", cs.kernel_name, timed(c2b, s, 3), NB, w * h * 3 // 2 + 4 * ow * oh)
    cs.close()


def staged():
    """kernel-only throughput of the gst-exact cells with YUV outputs (k_cs_staged_420 / _422), batched device frames"""
    s = torch.cuda.Stream()
    for (ifmt, w, h, ofmt, ow, oh, method) in [("NV12", 3840, 2160, "NV12", 1920, 1080, "bilinear"), ("NV12", 1920, 1080, "NV12", 1280, 720, "bilinear"),
                                               ("I420", 1920, 1080, "I420", 1280, 720, "bilinear"), ("BGRA", 1920, 1080, "NV12", 1920, 1080, "bilinear"),
                                               ("BGRA", 1920, 1080, "I420", 1280, 720, "bilinear"), ("BGRA", 1920, 1080, "UYVY", 1920, 1080, "bilinear"), ("NV12", 1920, 1080, "UYVY", 1920, 1080, "bilinear"),
                                               ("UYVY", 1920, 1080, "NV12", 1920, 1080, "bilinear"), ("UYVY", 1920, 1080, "UYVY", 1280, 720, "bilinear"),
                                               ("NV12", 3840, 2160, "NV12", 1920, 1080, "nearest"), ("NV12", 1920, 1080, "NV12", 1280, 720, "bicubic"),
                                               ("NV12", 3840, 2160, "NV12", 1920, 1080, "bicubic")]:
        isz, osz = vfhip.plane_layout(ifmt, w, h)[1], vfhip.plane_layout(ofmt, ow, oh)[1]
        N = max(16, int(1.2e9 // (isz + osz)))                     # > 1 GB per launch: several times the 256 MB infinity cache
        ip, op = (isz + 255) // 256 * 256, (osz + 255) // 256 * 256
        fin, fout = ring(N, isz, 70), ring(N, osz, 71)
        cs = vfhip.ConvertScale(0)
        cs.configure(ifmt, w, h, ofmt, ow, oh, method=method, colorimetry="bt709", chroma_site="mpeg2")

        def go():
            cs.process_device(fin.data_ptr(), fout.data_ptr(), stream=s.cuda_stream, n_frames=N, in_pitch=ip, out_pitch=op)
        report(f"{ifmt} {w}x{h} -> {ofmt} {ow}x{oh} {method} (gst-exact)", cs.kernel_name, timed(go, s, 5), N, isz + osz)
        cs.close()
        del fin, fout


def e2e():
    """element-level (PCIe-inclusive) rate of vfhip_convertscale_process on C2: pageable vs pinned host buffers"""
    import ctypes as C
    import time
    import numpy as np
    w, h, ow, oh = 3840, 2160, 1920, 1080
    in_size, out_size = vfhip.plane_layout("NV12", w, h)[1], 4 * ow * oh
    cs = vfhip.ConvertScale(0)
    cs.configure("NV12", w, h, "BGRA", ow, oh, colorimetry="bt2020", chroma_site="mpeg2")
    rng = np.random.default_rng(0)
    for kind in ("pageable", "pinned"):
        if kind == "pinned":
            pi, po = vfhip.lib.vfhip_pinned_alloc(0, in_size), vfhip.lib.vfhip_pinned_alloc(0, out_size)
            src = np.ctypeslib.as_array((C.c_uint8 * in_size).from_address(pi))
            dst = np.ctypeslib.as_array((C.c_uint8 * out_size).from_address(po))
        else:
            src, dst = np.empty(in_size, np.uint8), np.empty(out_size, np.uint8)
        src[:] = rng.integers(0, 256, in_size, dtype=np.uint8)
        fi = vfhip.frame_from_base(cs.in_info, "NV12", w, h, src.ctypes.data)
        fo = vfhip.frame_from_base(cs.out_info, "BGRA", ow, oh, dst.ctypes.data)
        for _ in range(5):
            vfhip.check(vfhip.lib.vfhip_convertscale_process(cs.h, C.byref(fi), C.byref(fo)))
        n, t0 = 100, time.perf_counter()
        for _ in range(n):
            vfhip.check(vfhip.lib.vfhip_convertscale_process(cs.h, C.byref(fi), C.byref(fo)))
        dt = time.perf_counter() - t0
        print(json.dumps({"config": f"C2 end-to-end vfhip_convertscale_process, {kind} host frames (sync per frame)",
                          "frames_per_s": round(n / dt, 1), "ms_per_frame": round(dt / n * 1e3, 3),
                          "pcie_GBps": round((in_size + out_size) * n / dt / 1e9, 2)}), flush=True)
        # the same frames two deep through submit / wait: upload of frame n+1 overlaps kernel + download of frame n
        n, t0 = 200, time.perf_counter()
        for k in range(n):
            vfhip.check(vfhip.lib.vfhip_convertscale_submit(cs.h, C.byref(fi), C.byref(fo)))
            if vfhip.lib.vfhip_convertscale_in_flight(cs.h) == 2:
                vfhip.check(vfhip.lib.vfhip_convertscale_wait(cs.h))
        while vfhip.lib.vfhip_convertscale_in_flight(cs.h):
            vfhip.check(vfhip.lib.vfhip_convertscale_wait(cs.h))
        dt = time.perf_counter() - t0
        print(json.dumps({"config": f"C2 end-to-end vfhip_convertscale_submit/_wait, {kind} host frames (two frames in flight)",
                          "frames_per_s": round(n / dt, 1), "ms_per_frame": round(dt / n * 1e3, 3),
                          "pcie_GBps": round((in_size + out_size) * n / dt / 1e9, 2)}), flush=True)
    cs.close()


def e2e_compositor():
    """C4 at the element level (PCIe-inclusive): 4 x BGRA 1080p + NV12 720p pinned host frames -> BGRA 2160p pinned host frame,
    synchronous vfhip_compositor_composite vs two composites in flight (vfhip_compositor_submit / _wait)"""
    import ctypes as C
    import time
    import numpy as np
    ow, oh = 3840, 2160
    comp = vfhip.Compositor(0).configure("BGRA", ow, oh, colorimetry="bt709")
    specs = [("BGRA", 1920, 1080, 0, 0), ("BGRA", 1920, 1080, 1920, 0), ("BGRA", 1920, 1080, 0, 1080), ("BGRA", 1920, 1080, 1920, 1080), ("NV12", 1280, 720, 1280, 720)]
    rng = np.random.default_rng(0)
    arr = (vfhip.PadInput * len(specs))()
    total = 0
    for i, (fmt, w, h, x, y) in enumerate(specs):
        size = vfhip.plane_layout(fmt, w, h)[1]
        total += size
        p = vfhip.lib.vfhip_pinned_alloc(0, size)
        np.ctypeslib.as_array((C.c_uint8 * size).from_address(p))[:] = rng.integers(0, 256, size, dtype=np.uint8)
        arr[i] = vfhip.Compositor.pad(fmt, w, h, p, x, y, w, h, 1.0 if i < 4 else 0.7, "over", "bt709")
    out_size = 4 * ow * oh
    total += out_size
    fo = vfhip.frame_from_base(comp.info, "BGRA", ow, oh, vfhip.lib.vfhip_pinned_alloc(0, out_size))
    lib, bg = vfhip.lib, vfhip.BACKGROUNDS["black"]
    for _ in range(3):
        vfhip.check(lib.vfhip_compositor_composite(comp.h, arr, len(specs), bg, C.byref(fo)))
    n, t0 = 60, time.perf_counter()
    for _ in range(n):
        vfhip.check(lib.vfhip_compositor_composite(comp.h, arr, len(specs), bg, C.byref(fo)))
    dt = time.perf_counter() - t0
    print(json.dumps({"config": "C4 end-to-end vfhip_compositor_composite, pinned host frames (sync per frame)", "frames_per_s": round(n / dt, 1),
                      "ms_per_frame": round(dt / n * 1e3, 3), "pcie_GBps": round(total * n / dt / 1e9, 2)}), flush=True)
    n, t0 = 120, time.perf_counter()
    for _ in range(n):
        vfhip.check(lib.vfhip_compositor_submit(comp.h, arr, len(specs), bg, C.byref(fo)))
        if lib.vfhip_compositor_in_flight(comp.h) == 2:
            vfhip.check(lib.vfhip_compositor_wait(comp.h))
    while lib.vfhip_compositor_in_flight(comp.h):
        vfhip.check(lib.vfhip_compositor_wait(comp.h))
    dt = time.perf_counter() - t0
    print(json.dumps({"config": "C4 end-to-end vfhip_compositor_submit/_wait, pinned host frames (two composites in flight)", "frames_per_s": round(n / dt, 1),
                      "ms_per_frame": round(dt / n * 1e3, 3), "pcie_GBps": round(total * n / dt / 1e9, 2)}), flush=True)
    comp.close()


def e2e_chain():
    """C5 chain at the element level (synchronous *_process calls, pinned host frames at both ends): deinterlace ->
    convertscale with the intermediate frame (a) in host memory — what a pipeline of system-memory elements does, four
    PCIe crossings — and (b) device-resident (memory:HIPMemory between the elements), two crossings"""
    import ctypes as C
    import time
    import numpy as np
    w, h, ow, oh = 3840, 2160, 1920, 1080
    in_size, out_size = vfhip.plane_layout("NV12", w, h)[1], 4 * ow * oh
    d = vfhip.Deinterlace(0)
    d.configure("NV12", w, h, colorimetry="bt709")
    cs = vfhip.ConvertScale(0)
    cs.configure("NV12", w, h, "BGRA", ow, oh, colorimetry="bt2020", chroma_site="mpeg2")
    pi, pm, po = (vfhip.lib.vfhip_pinned_alloc(0, n) for n in (in_size, in_size, out_size))
    src = np.ctypeslib.as_array((C.c_uint8 * in_size).from_address(pi))
    src[:] = np.random.default_rng(0).integers(0, 256, in_size, dtype=np.uint8)
    dev_mid = torch.zeros(in_size + 256, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    fi = vfhip.frame_from_base(d.info, "NV12", w, h, pi)
    fm_host = vfhip.frame_from_base(d.info, "NV12", w, h, pm)
    fm_dev = vfhip.frame_from_base(d.info, "NV12", w, h, dev_mid.data_ptr())
    fm_dev.flags |= 2                                   # VFHIP_FRAME_FLAG_DEVICE
    fo = vfhip.frame_from_base(cs.out_info, "BGRA", ow, oh, po)
    prm = vfhip.DeinterlaceParams(vfhip.DEINTERLACE_METHODS["greedyh"], 1, 0.1, 0)
    for kind, fm in (("host (system memory between the elements)", fm_host), ("device (memory:HIPMemory between the elements)", fm_dev)):
        def once():
            vfhip.check(vfhip.lib.vfhip_deinterlace_process(d.h, C.byref(fi), C.byref(fm), C.byref(prm)))
            vfhip.check(vfhip.lib.vfhip_convertscale_process(cs.h, C.byref(fm), C.byref(fo)))
        for _ in range(5):
            once()
        n, t0 = 100, time.perf_counter()
        for _ in range(n):
            once()
        dt = time.perf_counter() - t0
        print(json.dumps({"config": f"C5 chain end-to-end deinterlace greedyh -> convertscale NV12 2160p -> BGRA 1080p, intermediate frame: {kind}",
                          "frames_per_s": round(n / dt, 1), "ms_per_frame": round(dt / n * 1e3, 3)}), flush=True)
    d.close()
    cs.close()


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "main":
        main()
    elif len(sys.argv) > 1 and sys.argv[1] == "staged":
        staged()
    elif len(sys.argv) > 1 and sys.argv[1] == "e2e_compositor":
        e2e_compositor()
    elif len(sys.argv) > 1 and sys.argv[1] == "e2e":
        e2e()
        e2e_compositor()
        e2e_chain()
    else:
        main()
        e2e()
        e2e_chain()
