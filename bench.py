#!/usr/bin/env python3
"""bench.py — BASELINE.json headline: frames/s + achieved HBM GB/s, NV12 3840x2160 -> BGRA 1920x1080
bilinear (config[1]) through libvfhip's C ABI, gst-exact numerics, on N GPUs of one node.

A "step" is ONE batched launch of the hot-path kernel over a device-resident ring of `--frames`
distinct synthetic frames (inputs already in HBM; the ring — in + out — is far larger than the 256 MiB
Infinity Cache, so the traffic is real HBM traffic).  One process per GPU; frames/streams are
independent, so ranks share nothing (no collective on the data path; torch.distributed is used only
for the barrier and the max-over-ranks time) -> "scaling": "weak".

  python bench.py [--gpus N] [--steps K] [--warmup W] [--frames F]
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0 (see README / DESIGN.md §measurement for the fields).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "gstreamer-metal_amd"))

IN_W, IN_H, OUT_W, OUT_H = 3840, 2160, 1920, 1080
ALG_BYTES_PER_FRAME = IN_W * IN_H * 3 // 2 + OUT_W * OUT_H * 4      # 20,736,000
HBM_PEAK_GBS = 8000.0                                               # MI355X_MICROARCH.md: 8.0 TB/s spec


def cpu_baseline(seconds_budget=20.0):
    """The CPU oracle (bit-exact restatement of GStreamer 1.14 videoconvert+videoscale, OpenMP over the host
    cores) timed on a bounded sample of the same workload.  Reported next to the GPU number; not the target."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle_lib
    orc = oracle_lib.load()
    # the GPU box gives one GPU's share of the host: 16 cores (os.cpu_count() reports the whole machine)
    try:
        avail = len(os.sched_getaffinity(0))
    except Exception:
        avail = os.cpu_count() or 1
    cores = orc.lib.gst114_set_threads(min(avail, 16))
    rng = np.random.default_rng(0)
    raw = rng.integers(0, 256, IN_W * IN_H * 3 // 2, dtype=np.uint8)
    orc.convertscale("NV12", IN_W, IN_H, raw, "bt2020", "mpeg2", "bilinear", "BGRA", OUT_W, OUT_H)   # warm-up
    n, t0 = 0, time.perf_counter()
    while True:
        orc.convertscale("NV12", IN_W, IN_H, raw, "bt2020", "mpeg2", "bilinear", "BGRA", OUT_W, OUT_H)
        n += 1
        el = time.perf_counter() - t0
        if el > seconds_budget or n >= 400:
            break
    out = {"value": round(n / el, 2), "unit": "frames/s", "cores": cores, "kind": "port",
           "sample": f"{n} frames NV12 {IN_W}x{IN_H} -> BGRA {OUT_W}x{OUT_H}, oracle/gst114.c -O3 -march=native OpenMP x{cores}"}
    g = gstreamer_cpu_pipeline(cores)
    if g:
        out["gstreamer"] = g
    return out


def gstreamer_cpu_pipeline(threads, frames=48):
    """When the image's GStreamer 1.14 is present (it is on the GPU box), also time the real CPU elements on the same
    conversion: BASELINE configs[1] verbatim, wall clock minus a source-only run (SURVEY.md §8d CPU baseline)."""
    import gst_env                     # tests/gst_env.py: environment for /opt/conda's GStreamer (no oracle code)
    if not os.path.exists(gst_env.GST_LAUNCH):
        return None
    src = f"videotestsrc num-buffers={frames} ! video/x-raw,format=NV12,width={IN_W},height={IN_H}"
    res = {}
    try:
        gst_env.launch("videotestsrc num-buffers=1 ! videoconvert ! videoscale ! fakesink", timeout=120)   # registry scan, page-in
        t0 = time.perf_counter()
        if gst_env.launch(f"{src} ! fakesink sync=false", timeout=120).returncode != 0:
            return None
        t_src = time.perf_counter() - t0
        for label, n in (("1_thread", 1), (f"{threads}_threads", threads)):
            t0 = time.perf_counter()
            r = gst_env.launch(f"{src} ! videoconvert n-threads={n} ! videoscale n-threads={n} ! "
                               f"video/x-raw,format=BGRA,width={OUT_W},height={OUT_H} ! fakesink sync=false", timeout=300)
            if r.returncode != 0:
                return None
            dt = time.perf_counter() - t0 - t_src
            if dt <= 0:
                return None
            res[label] = round(frames / dt, 2)
    except Exception:
        return None
    return {"frames_per_s": res, "pipeline": "videotestsrc ! NV12 2160p ! videoconvert ! videoscale ! BGRA 1080p ! fakesink (GStreamer 1.14.0), source-only time subtracted",
            "sample": f"{frames} frames"}


def load_traffic(frames):
    """HBM bytes per launch from the committed PMC summary (separate rocprofv3 --pmc passes, corrected as
    MI355X_MICROARCH.md §HBM prescribes), rescaled to this run's frames per launch; None if absent."""
    p = os.path.join(ROOT, "profiles", "traffic_latest.json")
    try:
        with open(p) as f:
            t = json.load(f)
        return round(t["hbm_bytes_per_frame"] * frames)
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)   # ~0.3 s: long enough to sit at the sustained clock
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--frames", type=int, default=128, help="frames per launch (= ring size)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import bench_dist as bd
    import vfhip                                   # fails loudly when libvfhip.so is missing

    world, rank, local_rank = bd.world()
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    bd.init("nccl", torch.device("cuda", local_rank))          # barrier + max-over-ranks only; no data-path collective

    F = args.frames
    _, in_size = vfhip.plane_layout("NV12", IN_W, IN_H)
    in_pitch = (in_size + 255) // 256 * 256
    out_pitch = OUT_W * OUT_H * 4
    g = torch.Generator(device="cuda").manual_seed(0x9E3779B9 ^ rank)
    ring_in = torch.randint(0, 256, (F, in_pitch), dtype=torch.uint8, device="cuda", generator=g)   # synthetic, in HBM
    ring_out = torch.empty((F, out_pitch), dtype=torch.uint8, device="cuda")

    cs = vfhip.ConvertScale(local_rank)
    cs.configure("NV12", IN_W, IN_H, "BGRA", OUT_W, OUT_H, method="bilinear", numerics="gst-exact",
                 colorimetry="bt2020", chroma_site="mpeg2")          # GStreamer's default colorimetry at 2160 lines
    kernel = cs.kernel_name
    stream = torch.cuda.Stream()

    def step():
        cs.process_device(ring_in.data_ptr(), ring_out.data_ptr(), stream=stream.cuda_stream, n_frames=F,
                          in_pitch=in_pitch, out_pitch=out_pitch)

    def fence():
        bd.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(stream)                      # HIP events on the stream the kernel is launched on
    for _ in range(args.steps):
        step()
    ev1.record(stream)
    torch.cuda.synchronize()
    t_local = time.perf_counter() - t0
    fence()
    kernel_ms = ev0.elapsed_time(ev1) / args.steps          # average launch duration (back-to-back launches)

    t_max = bd.max_over_ranks(t_local, device="cuda")

    # the achievable ceiling next to the nominal one: a plain device-to-device copy of the input ring
    # (same size class as the job, far beyond the Infinity Cache), bytes read + bytes written per second
    copy_gbs = None
    if rank == 0:
        scratch = torch.empty_like(ring_in)
        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(stream):
            for _ in range(3):
                scratch.copy_(ring_in)
            c0.record(stream)
            for _ in range(30):
                scratch.copy_(ring_in)
            c1.record(stream)
        torch.cuda.synchronize()
        copy_gbs = 2 * ring_in.numel() * 30 / (c0.elapsed_time(c1) * 1e-3) / 1e9
        del scratch

    if rank == 0:
        fps = bd.whole_job_rate(F, args.steps, world, t_max)
        achieved = ALG_BYTES_PER_FRAME * F / (kernel_ms * 1e-3) / 1e9
        out = {
            "metric": "frames/sec + achieved HBM GB/s, NV12->BGRA 2160p->1080p",
            "value": round(fps, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(t_max / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "vfhipconvertscale NV12 3840x2160 -> BGRA 1920x1080 bilinear, gst-exact (BASELINE configs[1])",
                       "frames_per_step": F, "kernel": kernel, "colorimetry": "bt2020/mpeg2", "parallelism": f"independent-streams x{world}",
                       "device": vfhip.device_name(local_rank)},
            "achieved_GBps_whole_job": round(fps * ALG_BYTES_PER_FRAME / 1e9, 1),
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": load_traffic(F),
                         "kernel": kernel, "kernel_ms": round(kernel_ms, 4), "algorithmic_bytes_per_launch": ALG_BYTES_PER_FRAME * F,
                         "achievable": {"kind": f"device-to-device copy of the {ring_in.numel() / 1e9:.2f} GB input ring (read + write bytes)",
                                        "GBps": round(copy_gbs, 1), "frac": round(achieved / copy_gbs, 4)}},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    cs.close()
    bd.finish()


if __name__ == "__main__":
    main()
