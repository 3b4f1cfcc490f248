"""VFHIP_FRAME_FLAG_DEVICE: the synchronous entry points accept device-resident frames on either side (what the
plugin's `memory:HIPMemory` buffers carry) and produce the same bytes as with host frames."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.gpu
DEVICE = 2


def smooth(fmt, w, h, seed):
    from test_metal_elements_gpu import smooth as s
    return s(fmt, w, h, seed)


def dev_frame(vfhip, info, fmt, w, h, tensor):
    f = vfhip.frame_from_base(info, fmt, w, h, tensor.data_ptr())
    f.flags |= DEVICE
    return f


@pytest.mark.parametrize("din,dout", [(True, True), (True, False), (False, True)])
def test_convertscale_device_frames(vfhip, din, dout):
    import torch
    w, h, ow, oh = 256, 128, 128, 64
    raw = smooth("NV12", w, h, 1)
    cs = vfhip.ConvertScale(0)
    cs.configure("NV12", w, h, "BGRA", ow, oh, colorimetry="bt601", chroma_site="jpeg")
    want = cs.process(raw)
    tin = torch.from_numpy(raw).cuda()
    tout = torch.zeros(ow * oh * 4, dtype=torch.uint8, device="cuda")
    hout = np.zeros(ow * oh * 4, np.uint8)
    torch.cuda.synchronize()
    fi = dev_frame(vfhip, cs.in_info, "NV12", w, h, tin) if din else vfhip.frame_from_base(cs.in_info, "NV12", w, h, raw.ctypes.data)
    fo = dev_frame(vfhip, cs.out_info, "BGRA", ow, oh, tout) if dout else vfhip.frame_from_base(cs.out_info, "BGRA", ow, oh, hout.ctypes.data)
    vfhip.check(vfhip.lib.vfhip_convertscale_process(cs.h, C.byref(fi), C.byref(fo)))
    got = tout.cpu().numpy() if dout else hout            # complete on return: no extra synchronisation here
    assert np.array_equal(got.reshape(want.shape), want)
    cs.close()


def test_chain_stays_on_device(vfhip, metalref):
    """deinterlace -> convertscale -> videofilter -> transform with device frames in between == the same chain through
    host frames (what a gst-launch pipeline of vfhip elements negotiating memory:HIPMemory does)"""
    import torch
    w, h = 128, 72
    frames = [smooth("NV12", w, h, 10 + k) for k in range(3)]
    d, d2 = vfhip.Deinterlace(0), vfhip.Deinterlace(0)
    cs, vf, tr = vfhip.ConvertScale(0), vfhip.VideoFilter(0), vfhip.Transform(0)
    for x in (d, d2):
        x.configure("NV12", w, h)
    cs.configure("NV12", w, h, "RGBA", 64, 36, numerics="metal")
    vf.configure("RGBA", 64, 36)
    tr.configure("RGBA", 64, 36)
    prm = vfhip.filter_params(brightness=0.1, sharpness=0.4, gamma=1.3)
    dprm = vfhip.DeinterlaceParams(vfhip.DEINTERLACE_METHODS["greedyh"], 1, 0.05, 0)
    tprm = vfhip.TransformParams(vfhip.TRANSFORM_METHODS["horizontal-flip"], 0, 0, 0, 0)
    size = ol.raw_layout("NV12", w, h)[1]
    for raw in frames:
        # host chain
        a = d.process(raw, method="greedyh", tff=True, threshold=0.05)
        b = cs.process(a)
        c = vf.process(b, prm)
        want = tr.process(c, method="horizontal-flip")
        # device chain: one upload, one download
        t0 = torch.zeros(size, dtype=torch.uint8, device="cuda")
        t1 = torch.zeros(64 * 36 * 4, dtype=torch.uint8, device="cuda")
        t2 = torch.zeros(64 * 36 * 4, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        hin = vfhip.frame_from_base(d2.info, "NV12", w, h, raw.ctypes.data)
        f0 = dev_frame(vfhip, d2.info, "NV12", w, h, t0)
        vfhip.check(vfhip.lib.vfhip_deinterlace_process(d2.h, C.byref(hin), C.byref(f0), C.byref(dprm)))
        f1 = dev_frame(vfhip, cs.out_info, "RGBA", 64, 36, t1)
        vfhip.check(vfhip.lib.vfhip_convertscale_process(cs.h, C.byref(f0), C.byref(f1)))
        f2 = dev_frame(vfhip, vf.out_info, "RGBA", 64, 36, t2)
        vfhip.check(vfhip.lib.vfhip_videofilter_process(vf.h, C.byref(f1), C.byref(f2), C.byref(prm)))
        got = np.zeros(64 * 36 * 4, np.uint8)
        hout = vfhip.frame_from_base(tr.out_info, "RGBA", 64, 36, got.ctypes.data)
        vfhip.check(vfhip.lib.vfhip_transform_process(tr.h, C.byref(f2), C.byref(hout), C.byref(tprm)))
        assert np.array_equal(got, want)
    for x in (d, d2, cs, vf, tr):
        x.close()


def test_deinterlace_device_input_keeps_history(vfhip, metalref):
    import torch
    fmt, w, h = "I420", 64, 40
    size = ol.raw_layout(fmt, w, h)[1]
    frames = [smooth(fmt, w, h, 30 + k) for k in range(3)]
    d = vfhip.Deinterlace(0)
    d.configure(fmt, w, h)
    prm = vfhip.DeinterlaceParams(vfhip.DEINTERLACE_METHODS["weave"], 1, 0.1, 0)
    t = torch.zeros(size, dtype=torch.uint8, device="cuda")          # ONE input buffer, overwritten per frame (recycled pool buffer)
    prev = None
    for raw in frames:
        t.copy_(torch.from_numpy(raw))
        torch.cuda.synchronize()
        fi = dev_frame(vfhip, d.info, fmt, w, h, t)
        got = np.zeros(size, np.uint8)
        fo = vfhip.frame_from_base(d.info, fmt, w, h, got.ctypes.data)
        vfhip.check(vfhip.lib.vfhip_deinterlace_process(d.h, C.byref(fi), C.byref(fo), C.byref(prm)))
        want = metalref.deinterlace(fmt, w, h, raw, prev, 1, tff=True)
        assert np.abs(got.astype(int) - want.astype(int)).max() <= 1
        prev = raw
    d.close()


def test_compositor_device_pads_and_misaligned_device_frame(vfhip, metalref):
    import torch
    a, b = smooth("BGRA", 64, 48, 1), smooth("NV12", 32, 24, 2)
    comp = vfhip.Compositor(0)
    comp.configure("BGRA", 64, 48)
    want = comp.composite([("BGRA", 64, 48, a, 0, 0, 64, 48, 1.0, "over"), ("NV12", 32, 24, b, 10, 10, 32, 24, 0.5, "over")], background="black")
    ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    tout = torch.zeros(64 * 48 * 4 + 4, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    pa = comp.pad("BGRA", 64, 48, ta.data_ptr(), 0, 0, 64, 48, 1.0, "over")
    pb = comp.pad("NV12", 32, 24, tb.data_ptr(), 10, 10, 32, 24, 0.5, "over")
    pa.frame.flags |= DEVICE
    pb.frame.flags |= DEVICE
    arr = (vfhip.PadInput * 2)(pa, pb)
    fo = vfhip.frame_from_base(comp.info, "BGRA", 64, 48, tout.data_ptr())
    fo.flags |= DEVICE
    vfhip.check(vfhip.lib.vfhip_compositor_composite(comp.h, arr, 2, vfhip.BACKGROUNDS["black"], C.byref(fo)))
    assert np.array_equal(tout[:64 * 48 * 4].cpu().numpy(), want)
    bad = vfhip.frame_from_base(comp.info, "BGRA", 64, 48, tout.data_ptr() + 1)           # BGRA needs 4-byte alignment
    bad.flags |= DEVICE
    rc = vfhip.lib.vfhip_compositor_composite(comp.h, arr, 2, vfhip.BACKGROUNDS["black"], C.byref(bad))
    assert rc == -1 and b"aligned" in vfhip.lib.vfhip_last_error_string()
    comp.close()


def test_convertscale_submit_wait_pipeline(vfhip, oracle):
    """pipelined host path: frames submitted two deep come back in order and equal the synchronous result, for pageable
    and pinned buffers; misuse is reported, not crashed"""
    w, h, ow, oh, n = 256, 128, 128, 64, 7
    in_size = vfhip.plane_layout("NV12", w, h)[1]
    out_size = ow * oh * 4
    cs = vfhip.ConvertScale(0)
    cs.configure("NV12", w, h, "BGRA", ow, oh, colorimetry="bt601", chroma_site="jpeg")
    frames = [smooth("NV12", w, h, 50 + k) for k in range(n)]
    want = [cs.process(f) for f in frames]
    lib = vfhip.lib
    assert lib.vfhip_convertscale_wait(cs.h) == -1 and lib.vfhip_convertscale_in_flight(cs.h) == 0
    for kind in ("pageable", "pinned"):
        if kind == "pinned":
            ins = [np.ctypeslib.as_array((C.c_uint8 * in_size).from_address(lib.vfhip_pinned_alloc(0, in_size))) for _ in range(n)]
            outs = [np.ctypeslib.as_array((C.c_uint8 * out_size).from_address(lib.vfhip_pinned_alloc(0, out_size))) for _ in range(n)]
        else:
            ins, outs = [np.empty(in_size, np.uint8) for _ in range(n)], [np.empty(out_size, np.uint8) for _ in range(n)]
        for a, f in zip(ins, frames):
            a[:] = f
        for o in outs:
            o[:] = 0
        fi = [vfhip.frame_from_base(cs.in_info, "NV12", w, h, a.ctypes.data) for a in ins]
        fo = [vfhip.frame_from_base(cs.out_info, "BGRA", ow, oh, o.ctypes.data) for o in outs]
        done = 0
        for k in range(n):
            vfhip.check(lib.vfhip_convertscale_submit(cs.h, C.byref(fi[k]), C.byref(fo[k])))
            if lib.vfhip_convertscale_in_flight(cs.h) == 2:
                if k == 1:                                   # a third submit / a synchronous call with a full pipeline are refused
                    assert lib.vfhip_convertscale_submit(cs.h, C.byref(fi[k]), C.byref(fo[k])) == -1
                    assert lib.vfhip_convertscale_process(cs.h, C.byref(fi[k]), C.byref(fo[k])) == -1
                vfhip.check(lib.vfhip_convertscale_wait(cs.h))
                assert np.array_equal(outs[done].reshape(want[done].shape), want[done]), f"{kind} frame {done}"
                done += 1
        while lib.vfhip_convertscale_in_flight(cs.h):
            vfhip.check(lib.vfhip_convertscale_wait(cs.h))
            assert np.array_equal(outs[done].reshape(want[done].shape), want[done]), f"{kind} frame {done}"
            done += 1
        assert done == n
        if kind == "pinned":
            for a in ins + outs:
                lib.vfhip_pinned_free(a.ctypes.data)
    cs.close()


def test_many_handles_from_many_threads(vfhip, oracle, metalref):
    """SURVEY.md §8b threading: handles are single-caller, the device singleton is shared; eight threads, each with its own
    handles of a different element type / configuration, hammer the same GPU concurrently and every result is still exact
    (reference: tests/test-multi-element.sh, one command queue per renderer)"""
    import threading
    errors = []

    def worker(k):
        try:
            rng = np.random.default_rng(100 + k)
            if k % 4 == 0:
                w, h, ow, oh = 256, 128, 128, 64
                cs = vfhip.ConvertScale(0)
                cs.configure("NV12", w, h, "BGRA", ow, oh, colorimetry="bt709", chroma_site="mpeg2")
                for _ in range(12):
                    raw = rng.integers(0, 256, ol.raw_layout("NV12", w, h)[1], dtype=np.uint8)
                    want = oracle.convertscale("NV12", w, h, raw, "bt709", "mpeg2", "bilinear", "BGRA", ow, oh)
                    assert np.array_equal(cs.process(raw).reshape(oh, ow, 4), want)
                cs.close()
            elif k % 4 == 1:
                w, h = 96, 64
                vf = vfhip.VideoFilter(0)
                vf.configure("BGRA", w, h)
                prm = vfhip.filter_params(brightness=0.1, sharpness=0.5, gamma=1.4)
                for i in range(8):
                    raw = smooth("BGRA", w, h, 200 + 10 * k + i)
                    want = metalref.videofilter("BGRA", w, h, raw, "BGRA", ol.mr_filter_params(prm))
                    assert np.abs(vf.process(raw, prm).astype(int) - want.astype(int)).max() <= 1
                vf.close()
            elif k % 4 == 2:
                w, h = 128, 72
                d = vfhip.Deinterlace(0)
                d.configure("NV12", w, h)
                prev = None
                for i in range(8):
                    raw = smooth("NV12", w, h, 300 + 10 * k + i)
                    want = metalref.deinterlace("NV12", w, h, raw, prev, 3, tff=True, threshold=0.05)
                    assert np.abs(d.process(raw, method="greedyh", tff=True, threshold=0.05).astype(int) - want.astype(int)).max() <= 1
                    prev = raw
                d.close()
            else:
                w, h, ow, oh = 160, 90, 240, 135
                cs = vfhip.ConvertScale(0)
                cs.configure("I420", w, h, "RGBA", ow, oh, method="bicubic", colorimetry="bt601", chroma_site="jpeg")
                for _ in range(8):
                    raw = rng.integers(0, 256, ol.raw_layout("I420", w, h)[1], dtype=np.uint8)
                    want = oracle.convertscale("I420", w, h, raw, "bt601", "jpeg", "bicubic", "RGBA", ow, oh)
                    assert np.array_equal(cs.process(raw).reshape(oh, ow, 4), want)
                cs.close()
        except Exception as e:                        # noqa: BLE001
            errors.append((k, repr(e)))

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(8)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


# ---- padded strides / plane gaps / misaligned bases through every element's device-frame entry point ------------------
def _custom_layout(vfhip, fmt, w, h, pad, base):
    pl, _ = vfhip.plane_layout(fmt, w, h)
    out, off = [], base
    for (_, stride, rows) in pl:
        out.append((off, stride + pad, rows))
        off += (stride + pad) * rows + 5
    return out, off


def _repack(buf, layout_from, layout_to, size_to):
    out = np.zeros(size_to, np.uint8)
    for (fo, fs, rows), (to, ts, _) in zip(layout_from, layout_to):
        n = min(fs, ts)
        for r in range(rows):
            out[to + r * ts: to + r * ts + n] = buf[fo + r * fs: fo + r * fs + n]
    return out


def _meaningful(vfhip, fmt, w, h, raw):
    out = []
    for i, (off, stride, rows) in enumerate(vfhip.plane_layout(fmt, w, h)[0]):
        r = h if i == 0 else (h + 1) // 2
        wb = {"BGRA": 4 * w, "RGBA": 4 * w}.get(fmt, w if i == 0 else (w + 1) // 2 * (2 if fmt == "NV12" else 1))
        out.append(np.asarray(raw[off: off + r * stride]).reshape(r, stride)[:, :wb].reshape(-1))
    return np.concatenate(out)


@pytest.mark.parametrize("pad,base", [(16, 0), (3, 0), (16, 1), (5, 3)])
@pytest.mark.parametrize("ifmt,ofmt", [("BGRA", "BGRA"), ("NV12", "NV12"), ("I420", "BGRA"), ("RGBA", "I420")])
def test_elements_take_padded_and_misaligned_frames(vfhip, ifmt, ofmt, pad, base):
    """videofilter, deinterlace, transform, overlay and compositor on frames with padded / odd strides, gaps between planes and
    misaligned bases give the bytes they give on the default layout (the vector loads and stores all have byte-wise twins)"""
    import torch
    w, h = 70, 38
    rng = np.random.default_rng(pad * 10 + base)
    s = torch.cuda.Stream()

    def run(make, call, fin, fout, in_fmt=ifmt, out_fmt=ofmt):
        """call(elem, in_ptr, out_ptr, in_layout, out_layout) once on the default layout, once on the custom one"""
        res = []
        dpl_i, dsz_i = vfhip.plane_layout(in_fmt, w, h)
        dpl_o, dsz_o = vfhip.plane_layout(out_fmt, w, h)
        for custom in (False, True):
            (ipl, isz) = _custom_layout(vfhip, in_fmt, w, h, pad, base) if custom else (dpl_i, dsz_i)
            (opl, osz) = _custom_layout(vfhip, out_fmt, w, h, pad, base) if custom else (dpl_o, dsz_o)
            elem = make()
            outs = []
            for raw in fin:
                din = torch.from_numpy(_repack(raw, dpl_i, ipl, isz + 64)).cuda()
                dout = torch.zeros(osz + 64, dtype=torch.uint8, device="cuda")
                torch.cuda.synchronize()
                call(elem, din.data_ptr(), dout.data_ptr(), (ipl, isz), (opl, osz))
                s.synchronize()
                outs.append(_meaningful(vfhip, out_fmt, w, h, _repack(dout.cpu().numpy(), opl, dpl_o, dsz_o)))
            elem.close()
            res.append(outs)
        for a, b in zip(*res):
            assert np.array_equal(a, b), fout
    frames = [rng.integers(0, 256, vfhip.plane_layout(ifmt, w, h)[1], dtype=np.uint8) for _ in range(2)]

    prm = vfhip.filter_params(brightness=0.1, contrast=1.2, saturation=0.8, sharpness=0.5, vignette=0.3)
    run(lambda: vfhip.VideoFilter(0).configure(ifmt, w, h, ofmt, colorimetry="bt709"),
        lambda e, i, o, il, ol: e.process_device(i, o, prm, stream=s.cuda_stream, in_layout=il, out_layout=ol), frames, "videofilter")
    run(lambda: vfhip.Transform(0).configure(ifmt, w, h, ofmt, colorimetry="bt709"),
        lambda e, i, o, il, ol: e.process_device(i, o, method="rotate-180", crop=(2, 3, 4, 5), stream=s.cuda_stream, in_layout=il, out_layout=ol), frames, "transform")
    logo = rng.integers(0, 256, (9, 13, 4), dtype=np.uint8)

    def make_overlay():
        ov = vfhip.Overlay(0).configure(ifmt, w, h, ofmt, colorimetry="bt709")
        ov.set_image(logo)
        return ov
    run(make_overlay, lambda e, i, o, il, ol: e.process_device(i, o, x=5.0, y=7.0, alpha=0.6, stream=s.cuda_stream, in_layout=il, out_layout=ol), frames, "overlay")
    if ifmt == ofmt:
        run(lambda: vfhip.Deinterlace(0).configure(ifmt, w, h, colorimetry="bt709"),
            lambda e, i, o, il, ol: e.process_device(i, o, method="greedyh", tff=True, threshold=0.08, stream=s.cuda_stream, in_layout=il, out_layout=ol),
            frames, "deinterlace")

    def comp(e, i, o, il, ol):
        pads = [vfhip.Compositor.pad(ifmt, w, h, i, 3, 2, w, h, 0.8, "over", "bt709", layout=il), vfhip.Compositor.pad(ifmt, w, h, i, -11, 9, 40, 25, 0.5, "add", "bt709", layout=il)]
        e.composite_device(pads, o, background="checker", stream=s.cuda_stream, out_layout=ol)
    run(lambda: vfhip.Compositor(0).configure(ofmt, w, h, colorimetry="bt709"), comp, frames[:1], "compositor")


def test_compositor_submit_wait_pipeline(vfhip):
    """the compositor's pipelined host path: composites submitted two deep come back in order and equal the synchronous
    results (pads change from frame to frame); misuse is reported"""
    w, h, n = 160, 90, 6
    rng = np.random.default_rng(5)
    lib = vfhip.lib
    comp = vfhip.Compositor(0).configure("BGRA", w, h, colorimetry="bt709")
    a = [rng.integers(0, 256, vfhip.plane_layout("BGRA", 64, 48)[1], dtype=np.uint8) for _ in range(n)]
    b = [rng.integers(0, 256, vfhip.plane_layout("NV12", 80, 60)[1], dtype=np.uint8) for _ in range(n)]

    def pads(k):
        return [("BGRA", 64, 48, a[k], 5 + k, 3, 64, 48, 0.9, "over", "bt709"), ("NV12", 80, 60, b[k], 60, 20 - k, 100, 70, 0.6, "add", "bt709")]
    want = [comp.composite(pads(k), background="checker") for k in range(n)]
    outs = [np.zeros(w * h * 4, np.uint8) for _ in range(n)]
    keep, done = [], 0
    assert lib.vfhip_compositor_wait(comp.h) == -1 and lib.vfhip_compositor_in_flight(comp.h) == 0
    for k in range(n):
        arr = (vfhip.PadInput * 2)()
        for i, p in enumerate(pads(k)):
            arr[i] = vfhip.Compositor.pad(p[0], p[1], p[2], p[3].ctypes.data, *p[4:])
        fo = vfhip.frame_from_base(comp.info, "BGRA", w, h, outs[k].ctypes.data)
        keep.append((arr, fo))
        vfhip.check(lib.vfhip_compositor_submit(comp.h, arr, 2, vfhip.BACKGROUNDS["checker"], C.byref(fo)))
        if lib.vfhip_compositor_in_flight(comp.h) == 2:
            if k == 1:
                assert lib.vfhip_compositor_submit(comp.h, arr, 2, 0, C.byref(fo)) == -1
                assert lib.vfhip_compositor_composite(comp.h, arr, 2, 0, C.byref(fo)) == -1
            vfhip.check(lib.vfhip_compositor_wait(comp.h))
            assert np.array_equal(outs[done], want[done]), f"frame {done}"
            done += 1
    while lib.vfhip_compositor_in_flight(comp.h):
        vfhip.check(lib.vfhip_compositor_wait(comp.h))
        assert np.array_equal(outs[done], want[done]), f"frame {done}"
        done += 1
    assert done == n
    # cleanup with a frame still in flight must not crash
    vfhip.check(lib.vfhip_compositor_submit(comp.h, keep[0][0], 2, 0, C.byref(keep[0][1])))
    comp.close()


def test_compositor_pipelined_with_more_pads_than_one_pass_takes(vfhip):
    """20 pads (> 16 per kernel pass: the passes chain through the handle's two scratch targets) submitted two deep: the
    scratch targets are shared by the composites in flight and must be reused in stream order"""
    w, h, n, npads = 128, 72, 4, 20
    rng = np.random.default_rng(9)
    lib = vfhip.lib
    comp = vfhip.Compositor(0).configure("NV12", w, h, colorimetry="bt709")
    frames = [[rng.integers(0, 256, vfhip.plane_layout("BGRA", 24, 16)[1], dtype=np.uint8) for _ in range(npads)] for _ in range(n)]

    def pads(k):
        return [("BGRA", 24, 16, frames[k][i], (i * 11) % 100, (i * 7 + k) % 50, 30, 20, 0.4 + 0.03 * i, ["over", "add", "source"][i % 3], "bt709") for i in range(npads)]
    want = [comp.composite(pads(k), background="white") for k in range(n)]
    size = vfhip.plane_layout("NV12", w, h)[1]
    outs = [np.zeros(size, np.uint8) for _ in range(n)]
    keep, done = [], 0
    for k in range(n):
        arr = (vfhip.PadInput * npads)()
        for i, p in enumerate(pads(k)):
            arr[i] = vfhip.Compositor.pad(p[0], p[1], p[2], p[3].ctypes.data, *p[4:])
        fo = vfhip.frame_from_base(comp.info, "NV12", w, h, outs[k].ctypes.data)
        keep.append((arr, fo))
        vfhip.check(lib.vfhip_compositor_submit(comp.h, arr, npads, vfhip.BACKGROUNDS["white"], C.byref(fo)))
        if lib.vfhip_compositor_in_flight(comp.h) == 2:
            vfhip.check(lib.vfhip_compositor_wait(comp.h))
            assert np.array_equal(outs[done], want[done]), f"frame {done}"
            done += 1
    while lib.vfhip_compositor_in_flight(comp.h):
        vfhip.check(lib.vfhip_compositor_wait(comp.h))
        assert np.array_equal(outs[done], want[done]), f"frame {done}"
        done += 1
    assert done == n
    comp.close()
