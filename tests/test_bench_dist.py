"""The N>1 path of bench.py on CPU: world_size 2, gloo.  Frames / streams shard with no data-path collective; the
only communication is the barrier and the MAX of the elapsed times."""
import os
import subprocess
import sys
import textwrap
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import json, os, sys, time
    sys.path.insert(0, %r)
    import bench_dist as bd
    ws = bd.init("gloo")
    _, rank, _ = bd.world()
    bd.barrier()
    t0 = time.perf_counter()
    time.sleep(0.05 * (rank + 1))            # rank 1 is the slow one
    t = time.perf_counter() - t0
    bd.barrier()
    tmax = bd.max_over_ranks(t)
    assert tmax >= 0.1 - 1e-3, tmax          # every rank sees the slowest rank's time
    rate = bd.whole_job_rate(128, 4, ws, tmax)
    if rank == 0:
        print(json.dumps({"ws": ws, "tmax": tmax, "rate": rate, "shard": [bd.stream_to_gpu(s, ws) for s in range(5)]}))
    bd.finish()
""") % ROOT


def test_two_ranks_gloo(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29731", str(script)], capture_output=True, text=True, timeout=240, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    import json
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["ws"] == 2 and d["shard"] == [0, 1, 0, 1, 0]
    assert abs(d["rate"] - 2 * 128 * 4 / d["tmax"]) < 1e-6


def test_single_rank_is_a_noop():
    import bench_dist as bd
    assert bd.init("gloo") == 1
    bd.barrier()
    assert bd.max_over_ranks(1.5) == 1.5
    assert bd.whole_job_rate(10, 2, 1, 0.5) == 40


def _run_bench(args, env_extra=None, timeout=300):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, env=env)


def test_bench_py_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it: bench.py spawns 2 ranks itself (before touching a GPU), they meet
    on gloo, rank 0 prints ONE line with n_gpus = 2 and both devices listed.  --selftest-cpu replaces the GPU step by a sleep
    and marks the line as not-a-measurement (value null)."""
    import json
    r = _run_bench(["--gpus", "2", "--steps", "5", "--warmup", "1", "--selftest-cpu"])
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] is None and d["data"] == "selftest-no-gpu" and d["scaling"] == "weak"
    assert [x["rank"] for x in d["config"]["devices"]] == [0, 1]
    assert d["config"]["launches_per_step"] == 2                 # max over ranks of a per-rank value: the ranks really talked
    assert d["ms_per_step"] >= 4.0 - 0.5                          # the slow rank (4 ms per step) sets the time


def test_bench_py_eight_ranks_selftest():
    """the real world size of the node the driver scales to: `python bench.py --gpus 8 --selftest-cpu` spawns 8 ranks on 127.0.0.1, they rendezvous on
    gloo, pass the barriers, the device list is gathered from all eight in rank order, the step count is the MAX over ranks (8: rank 7's value) and
    the slowest rank (16 ms per step) sets the reported time.  No GPU, not a measurement: it shows the N = 8 path cannot hang or mis-aggregate."""
    import json
    r = _run_bench(["--gpus", "8", "--steps", "4", "--warmup", "1", "--selftest-cpu"], timeout=420)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                               # rank 0 alone prints
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["value"] is None and d["scaling"] == "weak"
    assert [x["rank"] for x in d["config"]["devices"]] == list(range(8)) and [x["ordinal"] for x in d["config"]["devices"]] == list(range(8))
    assert d["config"]["launches_per_step"] == 8
    assert d["ms_per_step"] >= 16.0 - 1.0


def test_bench_py_refuses_a_world_size_mismatch():
    r = _run_bench(["--gpus", "4", "--selftest-cpu"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29745"})
    assert r.returncode != 0 and "must agree" in (r.stdout + r.stderr)
    r = _run_bench(["--gpus", "1", "--selftest-cpu"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29746"})
    assert r.returncode != 0 and "must agree" in (r.stdout + r.stderr)


def test_bench_py_single_rank_selftest_and_traffic_key():
    import json
    r = _run_bench(["--selftest-cpu", "--steps", "3", "--warmup", "0"])
    assert r.returncode == 0, r.stdout + r.stderr
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 1
    sys.path.insert(0, ROOT)
    import bench
    t, note = bench.load_traffic(128)
    with open(os.path.join(ROOT, "profiles", "traffic_latest.json")) as f:
        committed = json.load(f)
    if committed.get("source_sha16") == bench.kernel_source_sha16():
        assert t == round(committed["hbm_bytes_per_frame"] * 128)
    else:
        assert t is None and "re-run" in note                    # a PMC figure of another kernel source is never printed


@pytest.mark.gpu
def test_bench_py_two_ranks_share_the_gpu_rehearsal():
    """The N-rank path on real HIP, on the one GPU a test box has: bench.py --gpus 2 --rehearse-shared-gpu, once self-launched and once under
    the driver's own launcher line.  Both ranks run the product path; the line says it is a rehearsal, not a scaling measurement."""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--gpus", "2", "--rehearse-shared-gpu", "--steps", "3", "--warmup", "1", "--frames", "64", "--no-ceilings", "--precondition", "0.1"]
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    for cmd in ([sys.executable, os.path.join(root, "bench.py")] + common,
                [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
                 os.path.join(root, "bench.py")] + common):
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=root)
        assert r.returncode == 0, r.stderr[-3000:]
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, r.stdout[-2000:]                  # rank 0 alone prints
        d = json.loads(lines[0])
        assert d["n_gpus"] == 2 and d["scaling"] == "weak" and "rehearsal" in d and d["value"] > 0
        assert [x["rank"] for x in d["config"]["devices"]] == [0, 1] and "others" not in d and "cpu_baseline" not in d
