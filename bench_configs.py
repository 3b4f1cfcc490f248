"""bench_configs.py — the BASELINE configs besides the headline (configs[0], [2], [3], [4]) as kernel-only, device-resident measurements with the headline's
protocol: a ring of distinct synthetic frames far beyond the 256 MiB Infinity Cache, batched launches through the C ABI, HIP events on the launch stream,
an untimed pre-conditioning phase, then >= 100 ms of back-to-back launches.  bench.py calls others () after its timed region and puts the result into
the `others` object of its one JSON line (rank 0, one GPU); tools/bench_elements.py prints the same figures line by line.  No oracle code here."""
import math
import time

PEAK_GBS = 8000.0


def _measure(torch, stream, fn, precondition_s=0.25, min_ms=100.0):
    for _ in range(2):
        fn()
    stream.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t_end = time.perf_counter() + precondition_s
    est = None
    while est is None or time.perf_counter() < t_end:
        e0.record(stream)
        for _ in range(4):
            fn()
        e1.record(stream)
        e1.synchronize()
        est = e0.elapsed_time(e1) / 4
    n = max(5, int(math.ceil(min_ms / est)))
    e0.record(stream)
    for _ in range(n):
        fn()
    e1.record(stream)
    e1.synchronize()
    return e0.elapsed_time(e1) / n, n


def _ring(torch, n, size, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return torch.randint(0, 256, (n, (size + 255) // 256 * 256), dtype=torch.uint8, device="cuda", generator=g)


def _entry(config, kernel, ms, launches, frames, alg_bytes, ring_bytes, **extra):
    gbs = alg_bytes * frames / (ms * 1e-3) / 1e9
    d = {"config": config, "frames_per_s": round(frames / ms * 1e3, 1), "kernel": kernel, "kernel_ms": round(ms, 4), "frames_per_launch": frames,
         "launches_timed": launches, "algorithmic_bytes_per_frame": alg_bytes, "achieved_GBps": round(gbs, 1), "frac": round(gbs / PEAK_GBS, 4),
         "ring_MiB": round(ring_bytes / 2**20)}
    d.update(extra)
    return d


def c1(torch, vfhip, stream, dev=0, frames=512):
    """BASELINE configs[0] on the GPU: NV12 1920x1080 -> BGRA 640x480, bilinear, gst-exact (the reference's CPU-runnable case)"""
    w, h, ow, oh = 1920, 1080, 640, 480
    size = vfhip.plane_layout("NV12", w, h)[1]
    fin, fout = _ring(torch, frames, size, 30), torch.empty((frames, 4 * ow * oh), dtype=torch.uint8, device="cuda")
    cs = vfhip.ConvertScale(dev)
    cs.configure("NV12", w, h, "BGRA", ow, oh, colorimetry="bt709", chroma_site="mpeg2")       # GStreamer's defaults at 1080 lines
    ms, n = _measure(torch, stream, lambda: cs.process_device(fin.data_ptr(), fout.data_ptr(), stream=stream.cuda_stream, n_frames=frames,
                                                              in_pitch=fin.shape[1], out_pitch=fout.shape[1]))
    # (the handle names the kernel FAMILY it configured, k_cs_taps; a launch of this size runs its four-rows-per-lane member, as the rocprofv3 stats show)
    kernel = "k_cs_taps_strip" if cs.kernel_name == "k_cs_taps" else cs.kernel_name
    out = _entry("configs[0]: vfhipconvertscale NV12 1920x1080 -> BGRA 640x480 bilinear, gst-exact", kernel, ms, n, frames, size + 4 * ow * oh,
                 fin.numel() + fout.numel())
    cs.close()
    return out


def c3_params(vfhip):
    """the reference's "all colour adjustments" set (tests/test-videofilter.sh:198-201) + invert + chroma key (:183-186) + a 33^3 LUT"""
    import numpy as np
    n = 33
    g = np.linspace(0, 1, n, dtype=np.float32)
    lut = np.ones((n, n, n, 4), np.float32)
    lut[..., 0], lut[..., 1], lut[..., 2] = g[None, None, :] ** 1.05, g[None, :, None], g[:, None, None] ** 0.95
    prm = vfhip.filter_params(brightness=0.1, contrast=1.2, saturation=0.8, hue=0.3 * math.pi, gamma=1.5, sharpness=0.5, sepia=0.2,
                              noise=0.1, vignette=0.3, invert=True, chroma_key=(0.0, 1.0, 0.0), tolerance=0.3, smoothness=0.1)
    return prm, lut


def c3(torch, vfhip, stream, dev=0, frames=64):
    """BASELINE configs[2]: vfhipvideofilter BGRA 1920x1080, all 15 properties + LUT, uniform random bytes (the LUT gather's worst case)"""
    w, h = 1920, 1080
    fin, fout = _ring(torch, frames, 4 * w * h, 1), torch.empty((frames, 4 * w * h), dtype=torch.uint8, device="cuda")
    vf = vfhip.VideoFilter(dev)
    vf.configure("BGRA", w, h)
    prm, lut = c3_params(vfhip)
    vf.set_lut(lut)
    ms, n = _measure(torch, stream, lambda: vf.process_device(fin.data_ptr(), fout.data_ptr(), prm, stream=stream.cuda_stream, n_frames=frames,
                                                              in_pitch=fin.shape[1], out_pitch=fout.shape[1]))
    out = _entry("configs[2]: vfhipvideofilter BGRA 1920x1080, all 15 properties + 33^3 LUT, single pass", "k_vf_sharp", ms, n, frames, 2 * 4 * w * h,
                 fin.numel() + fout.numel())
    # beside it, NOT the config's figure: the same launch on picture-like frames (smooth gradients + a little noise, every frame different) — neighbouring
    # pixels then fall into the same LUT cells, which is the case a grading filter meets in practice; uniform random bytes above are the gather's worst case
    yy = torch.arange(h, device="cuda", dtype=torch.float32).view(1, h, 1, 1)
    xx = torch.arange(w, device="cuda", dtype=torch.float32).view(1, 1, w, 1)
    ph = torch.arange(frames, device="cuda", dtype=torch.float32).view(frames, 1, 1, 1)
    ch = torch.tensor([0.0, 2.1, 4.2, 0.0], device="cuda").view(1, 1, 1, 4)
    img = 128 + 90 * torch.sin(xx / 211.0 + ph * 0.37 + ch) * torch.cos(yy / 173.0 + ch * 0.5) + torch.randint(-3, 4, (frames, h, w, 4), device="cuda")
    img[..., 3] = 255
    fin[:, :4 * w * h] = img.clamp_(0, 255).to(torch.uint8).view(frames, -1)
    del img
    ms2, n2 = _measure(torch, stream, lambda: vf.process_device(fin.data_ptr(), fout.data_ptr(), prm, stream=stream.cuda_stream, n_frames=frames,
                                                                in_pitch=fin.shape[1], out_pitch=fout.shape[1]), precondition_s=0.1, min_ms=60.0)
    out["picture_like_input"] = {"frames_per_s": round(frames / ms2 * 1e3, 1), "kernel_ms": round(ms2, 4), "frac": round(2 * 4 * w * h * frames / (ms2 * 1e-3) / 1e9 / PEAK_GBS, 4),
                                 "note": "smooth synthetic frames; beside the config's figure (uniform random bytes), not instead of it"}
    vf.close()
    return out


def c4(torch, vfhip, stream, dev=0, frames=32):
    """BASELINE configs[3]: vfhipcompositor 4 x BGRA 1080p quadrants (alpha .9, over) + NV12 720p centred (alpha .7) -> BGRA 2160p, black background"""
    ow, oh = 3840, 2160
    quads = [_ring(torch, frames, 4 * 1920 * 1080, 10 + k) for k in range(4)]
    nv = _ring(torch, frames, vfhip.plane_layout("NV12", 1280, 720)[1], 20)
    out = torch.empty((frames, 4 * ow * oh), dtype=torch.uint8, device="cuda")
    comp = vfhip.Compositor(dev)
    comp.configure("BGRA", ow, oh)
    pads = [comp.pad("BGRA", 1920, 1080, quads[q].data_ptr(), (q % 2) * 1920, (q // 2) * 1080, 1920, 1080, 0.9, "over") for q in range(4)]
    pads.append(comp.pad("NV12", 1280, 720, nv.data_ptr(), (ow - 1280) // 2, (oh - 720) // 2, 1280, 720, 0.7, "over", "bt709"))
    pitches = [quads[q].shape[1] for q in range(4)] + [nv.shape[1]]
    ms, n = _measure(torch, stream, lambda: comp.composite_device(pads, out.data_ptr(), background="black", stream=stream.cuda_stream, n_frames=frames,
                                                                  pad_pitches=pitches, out_pitch=out.shape[1]))
    alg = 4 * 4 * 1920 * 1080 + 1280 * 720 * 3 // 2 + 4 * ow * oh
    res = _entry("configs[3]: vfhipcompositor 4 x BGRA 1080p + 1 x NV12 720p, alpha / z-order blend -> BGRA 2160p", "k_compositor_quads + k_compositor_420", ms, n,
                 frames, alg, sum(q.numel() for q in quads) + nv.numel() + out.numel())
    comp.close()
    return res


def c5(torch, vfhip, stream, dev=0, frames=512):
    """BASELINE configs[4] per GPU: one stream, vfhipdeinterlace greedy-H NV12 2160p -> vfhipconvertscale BGRA 1080p, device-resident intermediate.
    512 frames per launch like the headline: its second leg IS the headline kernel at the headline's launch size, so that a rocprofv3 --stats of
    the whole bench.py command still shows one population of k_cs_nv12_half launches (its average must agree with roofline.kernel_ms)"""
    w, h, ow, oh = 3840, 2160, 1920, 1080
    size = vfhip.plane_layout("NV12", w, h)[1]
    fin, mid = _ring(torch, frames, size, 3), torch.empty((frames, (size + 255) // 256 * 256), dtype=torch.uint8, device="cuda")
    fout = torch.empty((frames, 4 * ow * oh), dtype=torch.uint8, device="cuda")
    # interlaced content: every second frame repeats its predecessor's bottom field (a static field: weave), the others move (bob) — ~50 % motion pixels
    y = fin[:, :w * h].view(frames, h, w)
    y[1::2, 1::2, :] = y[0:frames - frames % 2:2, 1::2, :]
    de = vfhip.Deinterlace(dev)
    de.configure("NV12", w, h)
    cs = vfhip.ConvertScale(dev)
    cs.configure("NV12", w, h, "BGRA", ow, oh, colorimetry="bt2020", chroma_site="mpeg2")
    sp, pitch = stream.cuda_stream, fin.shape[1]

    def leg_de():
        de.process_device(fin.data_ptr(), mid.data_ptr(), method="greedyh", tff=True, threshold=0.1, stream=sp, n_frames=frames, in_pitch=pitch, out_pitch=pitch)

    def leg_cs():
        cs.process_device(mid.data_ptr(), fout.data_ptr(), stream=sp, n_frames=frames, in_pitch=pitch, out_pitch=fout.shape[1])

    def chain():
        leg_de()
        leg_cs()
    ms, n = _measure(torch, stream, chain)
    de_ms, _ = _measure(torch, stream, leg_de, precondition_s=0.1, min_ms=50.0)
    cs_ms, _ = _measure(torch, stream, leg_cs, precondition_s=0.1, min_ms=50.0)
    alg_de, alg_cs = 3 * size, size + 4 * ow * oh
    res = _entry("configs[4] on one GPU: vfhipdeinterlace greedy-H NV12 3840x2160 -> vfhipconvertscale BGRA 1920x1080, one stream, device-resident intermediate",
                 "k_deinterlace_420q + " + cs.kernel_name, ms, n, frames, alg_de + alg_cs, fin.numel() + mid.numel() + fout.numel(),
                 legs={"k_deinterlace_420q": {"kernel_ms": round(de_ms, 4), "algorithmic_bytes_per_frame": alg_de, "frac": round(alg_de * frames / (de_ms * 1e-3) / 1e9 / PEAK_GBS, 4)},
                       cs.kernel_name: {"kernel_ms": round(cs_ms, 4), "algorithmic_bytes_per_frame": alg_cs, "frac": round(alg_cs * frames / (cs_ms * 1e-3) / 1e9 / PEAK_GBS, 4)}})
    de.close()
    cs.close()
    return res


def others(torch, vfhip, stream, dev=0):
    """every non-headline BASELINE config, one after the other (each frees its ring before the next starts); about 3 s in all"""
    out = {}
    for name, fn in (("C1", c1), ("C3", c3), ("C4", c4), ("C5", c5)):
        t0 = time.perf_counter()
        try:
            out[name] = fn(torch, vfhip, stream, dev)
            out[name]["wall_s"] = round(time.perf_counter() - t0, 2)
        except Exception as e:                       # a failing side measurement must not take the headline line with it; it is reported as what it is
            out[name] = {"error": f"{type(e).__name__}: {e}"}
        torch.cuda.empty_cache()
    return out
