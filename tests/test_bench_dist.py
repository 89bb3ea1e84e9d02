"""The N > 1 path of bench.py: one process per GPU, gloo for the barrier and the max-over-ranks only
(replicas, no data-path collective).  Runs on CPU with world_size 2 (--dry-run skips the GPU work)."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_ranks_gloo_max_reduce():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "40", "--warmup", "2",
           "--dry-run"]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1  # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak"
    # rank 1 sleeps 0.10 s, rank 0 0.05 s: the reported time is the slower rank's, the value the whole job's
    assert 0.095 < d["elapsed_s"] < 0.5
    assert abs(d["value"] - 2 * 40 / d["elapsed_s"]) < 1e-6


def test_single_process_dry_run():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run", "--steps", "10"], cwd=ROOT,
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 1


def test_cpu_affinity_of_the_bench():
    """bench.pin_to_device: the list parser, the cases in which nothing is pinned, and — where this machine's sysfs names NUMA node 0 —
    a pin that stays inside the allowed set and is undone afterwards"""
    sys.path.insert(0, ROOT)
    import bench

    assert bench._cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    assert bench._cpulist("") == []

    class Pkg:
        def __init__(self, node):
            self.node = node

        def device_numa_node(self, device):
            return self.node

    assert bench.pin_to_device(Pkg(0), 0, "none")["mode"] == "none"
    assert bench.pin_to_device(Pkg(-1), 0, "ccx")["mode"] == "none"            # the device's node is unknown: nothing is touched
    if not hasattr(os, "sched_getaffinity") or not os.path.exists("/sys/devices/system/node/node0/cpulist"):
        return
    before = os.sched_getaffinity(0)
    try:
        for mode, rank in (("node", None), ("ccx", None), ("ccx", 1)):
            info = bench.pin_to_device(Pkg(0), 0, mode, rank)
            now = os.sched_getaffinity(0)
            assert now <= before and len(now) >= 1
            if info["mode"] != "none":
                assert info["cpus"] == len(now)
            os.sched_setaffinity(0, before)
    finally:
        os.sched_setaffinity(0, before)
