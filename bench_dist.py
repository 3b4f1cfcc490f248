"""bench_dist.py — the rank coordination bench.py uses for N GPUs of one node (one process per GPU, independent
streams, no data-path collective): barrier before/after the timed region, MAX over ranks of the elapsed time, rank 0
reports whole-job throughput.  Backend-agnostic so that the same code runs under `gloo` on CPU in the tests."""
import os

import torch
import torch.distributed as dist


def world():
    return int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))


def init(backend, device=None):
    ws, _, _ = world()
    if ws > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
        dist.init_process_group(backend, **kw)
    return ws


def barrier():
    if dist.is_initialized():
        dist.barrier()


def max_over_ranks(seconds, device="cpu"):
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    if dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_objects(obj):
    """every rank's small python object, in rank order (device list for the report)"""
    if not dist.is_initialized():
        return [obj]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, obj)
    return out


def stream_to_gpu(stream_index, n_gpus):
    """independent streams shard round-robin: stream s -> GPU s mod N"""
    return stream_index % n_gpus


def whole_job_rate(units_per_rank_per_step, steps, world_size, t_max):
    return world_size * units_per_rank_per_step * steps / t_max


def finish():
    if dist.is_initialized():
        dist.destroy_process_group()
