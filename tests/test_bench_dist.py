"""The N>1 path of bench.py on CPU: world_size 2, gloo.  Frames / streams shard with no data-path collective; the
only communication is the barrier and the MAX of the elapsed times."""
import os
import subprocess
import sys
import textwrap

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
