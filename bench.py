#!/usr/bin/env python3
"""bench.py — BASELINE.json headline: frames/s + achieved HBM GB/s of `vfhipconvertscale` NV12 3840x2160 -> BGRA
1920x1080 bilinear (configs[1]) through libvfhip's C ABI, gst-exact numerics, on N GPUs of one node.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--frames F] [--workload c2|c5]

What a "step" is: L back-to-back batched launches of the hot-path kernel over a device-resident ring of F distinct
synthetic frames (inputs already in HBM; ring in + out = F x 20.7 MB, far beyond the 256 MiB Infinity Cache), L chosen
so that a step lasts >= ~10 ms (`config.launches_per_step`, `config.frames_per_step` = L x F) — a 0.5 ms step would make
the driver's `--steps 20 --warmup 5` a 13 ms measurement taken on the boost clock.  Before the counted warm-up the
kernel runs untimed for `--precondition` seconds (default 0.5) so that the timed region sits on the sustained clock;
sclk / socket power sampled from sysfs during that phase are printed next to the result (`clocks`).

N GPUs: one process per GPU, stream s -> GPU s mod N, every rank owns its ring; frames / streams are independent, so
there is NO collective on the data path and RCCL is not initialised at all: ranks meet on a CPU `gloo` group for the
barrier on both sides of the timed region and for the MAX of the elapsed times -> "scaling": "weak".  With `--gpus N`
and no WORLD_SIZE in the environment this script starts the N ranks itself (torch.distributed.run on 127.0.0.1) BEFORE
anything touches a GPU; under an external launcher WORLD_SIZE must equal --gpus.

`--workload c5` = BASELINE configs[4] per GPU: `vfhipdeinterlace method=greedyh` (NV12 2160p) -> `vfhipconvertscale`
(BGRA 1080p), intermediate frames device-resident, both legs batched.

At one GPU the line also carries `others`: BASELINE configs[0], [2], [3] and [4] (C1, C3, C4, the C5 chain) measured right after the headline's
timed region with the same protocol (bench_configs.py) — frames/s, kernel, ms per launch, algorithmic bytes, fraction of 8 TB/s, ring size.

Prints ONE JSON line on rank 0 (fields: README.md / DESIGN.md §6).
"""
import argparse
import glob
import hashlib
import json
import os
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "gstreamer-metal_amd")
sys.path.insert(0, PKG)

IN_W, IN_H, OUT_W, OUT_H = 3840, 2160, 1920, 1080
NV12_BYTES = IN_W * IN_H * 3 // 2
ALG_C2 = NV12_BYTES + OUT_W * OUT_H * 4                # 20,736,000 B per frame
ALG_DEINT = 3 * NV12_BYTES                             # cur + prev read, out written: 37,324,800 B per frame
HBM_PEAK_GBS = 8000.0                                  # MI355X_MICROARCH.md: 8.0 TB/s spec
METRIC = "frames/sec + achieved HBM GB/s, NV12->BGRA 2160p->1080p"


# ------------------------------------------------------------------------------------------------ helpers (no GPU)
def kernel_source_sha16():
    """identity of the headline kernel's source: the committed PMC traffic figure is only valid for this exact code"""
    h = hashlib.sha256()
    for f in ("csrc/convertscale_kernels.h", "csrc/convertscale.hip"):
        with open(os.path.join(PKG, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def csrc_sha16():
    """identity of ALL kernel sources (the VALU figures of profiles/valu_latest.json cover every element kernel)"""
    h = hashlib.sha256()
    d = os.path.join(PKG, "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            with open(os.path.join(d, f), "rb") as fh:
                h.update(fh.read())
    return h.hexdigest()[:16]


def load_valu():
    """per-kernel VALU roofline figures (tools/valu_roofline.py on committed PMC passes): kernel -> {valu_issue_share, avg_issue_cycles}, only when they
    were measured on exactly these sources; else {} — a VALU-bound kernel's HBM fraction says how far it is from the wrong roof, this says how near the right one"""
    try:
        with open(os.path.join(ROOT, "profiles", "valu_latest.json")) as f:
            v = json.load(f)
    except Exception:
        return {}
    if v.get("source_sha16") != csrc_sha16():
        return {}
    return {k: {"valu_issue_share": d["valu_issue_share"], "avg_issue_cycles": d["avg_issue_cycles"], "source": "profiles/valu_latest.json (PMC SQ_INSTS_VALU x opcode-mix issue cost / SIMD cycles; lower bound)"}
            for k, d in v.get("kernels", {}).items()}


def load_traffic(frames):
    """HBM bytes per launch from the committed PMC summary (separate rocprofv3 --pmc passes, FETCH_SIZE x2 per the gfx950
    correction + WRITE_SIZE: MI355X_MICROARCH.md §HBM), rescaled to this run's frames per launch.  The summary names the
    sha of the kernel source it was measured on; for any other source the figure is stale and NOT printed (None)."""
    p = os.path.join(ROOT, "profiles", "traffic_latest.json")
    try:
        with open(p) as f:
            t = json.load(f)
    except Exception:
        return None, "no committed PMC summary"
    if t.get("source_sha16") != kernel_source_sha16():
        return None, f"committed PMC summary is for kernel source {t.get('source_sha16')}, this is {kernel_source_sha16()}: re-run tools/gpu_pmc.sh"
    return round(t["hbm_bytes_per_frame"] * frames), t.get("source", "")


class ClockSampler(threading.Thread):
    """sclk (MHz) and socket power (W) from sysfs hwmon while the kernel runs; the busiest card is reported"""

    def __init__(self, period=0.05):
        super().__init__(daemon=True)
        self.period, self.stop_flag, self.samples = period, False, {}
        self.cards = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"))

    @staticmethod
    def _read(path):
        try:
            with open(path) as f:
                return float(f.read().strip())
        except Exception:
            return None

    def run(self):
        while not self.stop_flag:
            for c in self.cards:
                f = self._read(os.path.join(c, "freq1_input"))
                p = self._read(os.path.join(c, "power1_average")) or self._read(os.path.join(c, "power1_input"))
                if f is not None or p is not None:
                    self.samples.setdefault(c, []).append((f, p))
            time.sleep(self.period)

    def result(self):
        self.stop_flag = True
        best = None
        for c, v in self.samples.items():
            fs = [a for a, _ in v if a]
            ps = [b for _, b in v if b]
            pw = sum(ps) / len(ps) / 1e6 if ps else 0.0
            if best is None or pw > best["socket_power_W"]:
                best = {"sclk_MHz": round(sum(fs) / len(fs) / 1e6) if fs else None, "socket_power_W": round(pw, 1), "samples": len(v),
                        "source": "sysfs hwmon during the pre-conditioning phase"}
        return best


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def cpu_baseline(workload, seconds_budget=20.0):
    """The CPU oracle timed on a bounded sample of the same workload on this box's host cores (rank 0, N = 1 only).
    Reported next to the GPU number; a baseline, not the target."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle_lib
    orc = oracle_lib.load()
    try:
        avail = len(os.sched_getaffinity(0))
    except Exception:
        avail = os.cpu_count() or 1
    cores = orc.lib.gst114_set_threads(min(avail, 16))       # one GPU's share of the host: 16 cores
    rng = np.random.default_rng(0)
    raw = rng.integers(0, 256, NV12_BYTES, dtype=np.uint8)
    conv = lambda x: orc.convertscale("NV12", IN_W, IN_H, x, "bt2020", "mpeg2", "bilinear", "BGRA", OUT_W, OUT_H)
    if workload == "c5":
        mr = oracle_lib.load_metalref()
        prev = rng.integers(0, 256, NV12_BYTES, dtype=np.uint8)
        n, t0 = 0, time.perf_counter()
        while True:
            conv(mr.deinterlace("NV12", IN_W, IN_H, raw, prev, 3, tff=True, threshold=0.1))
            n += 1
            el = time.perf_counter() - t0
            if el > seconds_budget or n >= 16:
                break
        return {"value": round(n / el, 2), "unit": "frames/s", "cores": cores, "kind": "port",
                "sample": f"{n} frames: oracle/metalref.c greedy-H deinterlace NV12 {IN_W}x{IN_H} (1 thread) -> oracle/gst114.c convert+scale to BGRA {OUT_W}x{OUT_H} (OpenMP x{cores})"}
    conv(raw)
    n, t0 = 0, time.perf_counter()
    while True:
        conv(raw)
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


def stream_ceilings(torch, stream, in_buf, out_buf):
    """what this box's HBM gives to the simplest streaming kernels of the job's size class (libvfhip_bench.so, not part
    of the product library): a one-access-per-lane copy and a flat kernel with the job's 3:2 read:write ratio"""
    import ctypes as C
    path = os.path.join(PKG, "libvfhip_bench.so")
    if not os.path.exists(path):
        return None
    lib = C.CDLL(path)
    lib.vfhip_bench_stream.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.c_void_p,
                                       C.POINTER(C.c_double), C.POINTER(C.c_double)]
    out = {}
    nin, nout = min(in_buf.numel(), 3 << 30), min(out_buf.numel(), 2 << 30)
    for kind, name in ((0, "copy_1to1"), (1, "mix_3to2_nt"), (2, "read_only_nt"), (3, "write_only")):
        ms, by = C.c_double(), C.c_double()
        rc = lib.vfhip_bench_stream(kind, in_buf.data_ptr(), out_buf.data_ptr(), nin, nout, 5, 40, stream.cuda_stream, C.byref(ms), C.byref(by))
        if rc == 0 and ms.value > 0:
            out[name] = round(by.value / (ms.value * 1e-3) / 1e9, 1)
    return out or None


# ------------------------------------------------------------------------------------------------ main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--frames", type=int, default=0, help="frames per launch (= ring size); default 512 (c2) / 128 (c5)")
    ap.add_argument("--launches-per-step", type=int, default=0, help="default: as many as make a step last >= --min-step-ms")
    ap.add_argument("--min-step-ms", type=float, default=10.0)
    ap.add_argument("--precondition", type=float, default=0.5, help="seconds of untimed launches before the counted warm-up")
    ap.add_argument("--workload", choices=("c2", "c5"), default="c2")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ceilings", action="store_true")
    ap.add_argument("--no-others", action="store_true", help="skip the other BASELINE configs (C1, C3, C4, C5) measured after the headline's timed region")
    ap.add_argument("--rehearse-shared-gpu", action="store_true",
                    help="let the N ranks share the visible GPUs (ordinal = LOCAL_RANK %% device count): a rehearsal of the N-rank path on a box with fewer GPUs; the line is marked, it is not a scaling measurement")
    ap.add_argument("--selftest-cpu", action="store_true",
                    help="rank plumbing only (spawn, gloo rendezvous, barrier, max over ranks, aggregation) with a sleep in place of the GPU step; prints a line marked as such, never a measurement")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")

    # ---- N > 1 without a launcher: start the N ranks ourselves, before torch / HIP are imported in this process
    env_world = os.environ.get("WORLD_SIZE")
    if args.gpus > 1 and env_world is None:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        raise SystemExit(subprocess.run(cmd, env=env).returncode)
    world = int(env_world or "1")
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: they must agree (one rank per GPU)")
    rank, local_rank = int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))

    import torch
    import bench_dist as bd
    bd.init("gloo")                         # CPU group: barrier + max over ranks only.  RCCL is never initialised.

    if args.selftest_cpu:
        return selftest_cpu(args, bd, world, rank)

    import vfhip                                   # fails loudly when libvfhip.so is missing
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    if local_rank >= torch.cuda.device_count() and not args.rehearse_shared_gpu:
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local_rank} but only {torch.cuda.device_count()} device(s) visible")
    if args.rehearse_shared_gpu:
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev_name = vfhip.device_name(local_rank)
    devices = bd.gather_objects({"rank": rank, "ordinal": local_rank, "name": dev_name})

    c5 = args.workload == "c5"
    F = args.frames or (128 if c5 else 512)
    in_pitch = (NV12_BYTES + 255) // 256 * 256
    out_pitch = OUT_W * OUT_H * 4
    g = torch.Generator(device="cuda").manual_seed(0x9E3779B9 ^ rank)
    ring_in = torch.randint(0, 256, (F, in_pitch), dtype=torch.uint8, device="cuda", generator=g)   # synthetic, in HBM
    ring_out = torch.empty((F, out_pitch), dtype=torch.uint8, device="cuda")
    ring_mid = torch.empty((F, in_pitch), dtype=torch.uint8, device="cuda") if c5 else None
    if c5:
        # interlaced content: two different random frames woven by line parity = ~50 % motion pixels for greedy-H
        # every second frame repeats its predecessor's bottom field (a static field: weave), the others move (bob)
        y = ring_in[:, :IN_W * IN_H].view(F, IN_H, IN_W)
        y[1::2, 1::2, :] = y[0:F - F % 2:2, 1::2, :]

    cs = vfhip.ConvertScale(local_rank)
    cs.configure("NV12", IN_W, IN_H, "BGRA", OUT_W, OUT_H, method="bilinear", numerics="gst-exact",
                 colorimetry="bt2020", chroma_site="mpeg2")          # GStreamer's default colorimetry at 2160 lines
    kernel = cs.kernel_name
    assert cs.numerics_in_effect == "gst-exact"
    de = None
    if c5:
        de = vfhip.Deinterlace(local_rank)
        de.configure("NV12", IN_W, IN_H)
    stream = torch.cuda.Stream()
    sp = stream.cuda_stream

    def launch_cs(src):
        cs.process_device(src.data_ptr(), ring_out.data_ptr(), stream=sp, n_frames=F, in_pitch=in_pitch, out_pitch=out_pitch)

    def launch_de():
        de.process_device(ring_in.data_ptr(), ring_mid.data_ptr(), method="greedyh", tff=True, threshold=0.1, stream=sp,
                          n_frames=F, in_pitch=in_pitch, out_pitch=in_pitch)

    def launch():
        if c5:
            launch_de()
            launch_cs(ring_mid)
        else:
            launch_cs(ring_in)

    def timed_launches(n, fn=launch):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)                     # HIP events on the stream the kernels are launched on
        for _ in range(n):
            fn()
        e1.record(stream)
        e1.synchronize()
        return e0.elapsed_time(e1) / n

    # ---- untimed pre-conditioning: sustained clock, page tables and caches in their steady state
    sampler = ClockSampler()
    sampler.start()
    t_end = time.perf_counter() + max(args.precondition, 0.05)
    est_ms = timed_launches(2)
    while time.perf_counter() < t_end:
        est_ms = timed_launches(8)
    clocks = sampler.result()
    L = args.launches_per_step or max(1, int(-(-args.min_step_ms // est_ms)))
    L = int(bd.max_over_ranks(float(L)))       # the same step on every rank

    def step():
        for _ in range(L):
            launch()

    def fence():
        torch.cuda.synchronize()
        bd.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(stream)
    for _ in range(args.steps):
        step()
    ev1.record(stream)
    torch.cuda.synchronize()
    t_local = time.perf_counter() - t0
    fence()
    launch_ms = ev0.elapsed_time(ev1) / (args.steps * L)     # average duration of one launch (c5: one pair), back to back
    t_max = bd.max_over_ranks(t_local)

    out = None
    if rank == 0:
        frames_per_step = L * F
        fps = bd.whole_job_rate(frames_per_step, args.steps, world, t_max)
        if c5:
            de_ms, cs_ms = timed_launches(5, launch_de), timed_launches(5, lambda: launch_cs(ring_mid))
            dom, dom_ms, dom_bytes = ("k_deinterlace_420q", de_ms, ALG_DEINT * F) if de_ms >= cs_ms else (kernel, cs_ms, ALG_C2 * F)
            alg_frame = ALG_DEINT + ALG_C2
            workload = ("vfhipdeinterlace greedy-H NV12 3840x2160 -> vfhipconvertscale BGRA 1920x1080 bilinear gst-exact, one stream per GPU, "
                        "device-resident intermediate (BASELINE configs[4])")
            traffic, traffic_note = None, "no PMC summary committed for this workload"
        else:
            dom, dom_ms, dom_bytes = kernel, launch_ms, ALG_C2 * F
            alg_frame = ALG_C2
            workload = "vfhipconvertscale NV12 3840x2160 -> BGRA 1920x1080 bilinear, gst-exact (BASELINE configs[1])"
            traffic, traffic_note = load_traffic(F)
        achieved = dom_bytes / (dom_ms * 1e-3) / 1e9
        roof = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": traffic, "traffic_source": traffic_note, "kernel": dom, "kernel_ms": round(dom_ms, 4),
                "algorithmic_bytes_per_launch": dom_bytes, "kernel_source_sha16": kernel_source_sha16()}
        valu = load_valu()
        if dom in valu:
            roof["valu"] = valu[dom]
        if c5:
            roof["legs_ms_per_launch"] = {"k_deinterlace_420q": round(de_ms, 4), kernel: round(cs_ms, 4)}
        if not args.no_ceilings:
            ceil = stream_ceilings(torch, stream, ring_in, ring_out)
            if ceil:
                ref = ceil.get("mix_3to2_nt") or ceil.get("copy_1to1")
                roof["achievable"] = {"kind": "streaming kernels of libvfhip_bench.so on this box, same buffers (GB/s, bytes read + written): one 16-byte access pair per lane; "
                                              "mix_3to2_nt has the job's read:write ratio", **ceil, "frac_of_mix_3to2": round(achieved / ref, 4)}
        out = {
            "metric": METRIC, "value": round(fps, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(t_max / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": workload, "frames_per_launch": F, "launches_per_step": L, "frames_per_step": frames_per_step,
                       "kernel": kernel, "colorimetry": "bt2020/mpeg2", "parallelism": f"independent-streams x{world}",
                       "rendezvous": "gloo (barrier + max over ranks); no data-path collective, RCCL not initialised",
                       "precondition_s": args.precondition, "devices": devices},
            "achieved_GBps_whole_job": round(fps * alg_frame / 1e9, 1),
            "launch_ms": round(launch_ms, 4),
            "clocks": clocks,
            "roofline": roof,
        }
        if args.rehearse_shared_gpu:
            out["rehearsal"] = f"{world} ranks on {torch.cuda.device_count()} GPU(s) (--rehearse-shared-gpu): the N-rank path exercised on real HIP, NOT a scaling measurement"
        if world == 1 and not args.no_others and not c5:
            # the other BASELINE configs, kernel-only like the headline, AFTER its timed region and with its rings released (bench_configs.py)
            import bench_configs
            del ring_in, ring_out
            torch.cuda.empty_cache()
            out["others"] = bench_configs.others(torch, vfhip, stream, local_rank)
            for o in out["others"].values():               # the VALU figure of the config's dominant kernel, when one is committed for these sources
                k0 = str(o.get("kernel", "")).split(" + ")[0]
                if k0 in valu:
                    o["valu"] = valu[k0]
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.workload)
        print(json.dumps(out), flush=True)
    cs.close()
    if de:
        de.close()
    bd.finish()
    return out


def selftest_cpu(args, bd, world, rank):
    """no GPU: the rank plumbing of the N-GPU path (used by tests/test_bench_dist.py with 2 ranks)"""
    devices = bd.gather_objects({"rank": rank, "ordinal": int(os.environ.get("LOCAL_RANK", "0")), "name": "none (selftest)"})
    L = int(bd.max_over_ranks(float(1 + rank)))

    def step():
        time.sleep(0.002 * (rank + 1))             # the last rank is the slow one
    for _ in range(args.warmup):
        step()
    bd.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    t_local = time.perf_counter() - t0
    bd.barrier()
    t_max = bd.max_over_ranks(t_local)
    if rank == 0:
        print(json.dumps({"metric": METRIC, "value": None, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(t_max / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                          "data": "selftest-no-gpu", "config": {"workload": "rank plumbing self-test (no GPU work, not a measurement)",
                                                                 "launches_per_step": L, "devices": devices}}), flush=True)
    bd.finish()


if __name__ == "__main__":
    main()
