"""The N > 1 path of bench.py end to end on the one-GPU box: two ranks share cuda:0 and exchange
through the library's callback communicator over gloo (PCGX_BENCH_REHEARSE=1; RCCL refuses two ranks
on one device).  Checks what the driver's multi-GPU run relies on: launch (under torchrun, and by
the script itself), tile construction, pcgx_icp_session_step_sharded on every rank, max-over-ranks
timing, ONE JSON line from rank 0 with whole-job units."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(cmd, env):
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=280, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1  # rank 0 only
    return json.loads(lines[0])


def _check(d):
    assert d["n_gpus"] == 2 and d["steps"] == 8 and d["warmup"] == 4 and d["scaling"] == "weak"
    assert d["config"]["target_points_total"] == 200000 and "x2" in d["config"]["parallelism"]
    assert abs(d["value"] - 200000 / (d["ms_per_step"] * 1e-3) / 1e6) < 1e-6 * d["value"]
    assert d["final_value"] == d["final_value"] and d["final_value"] < 1e-3  # the sharded Fit converges
    # the headline of an N > 1 run is in the library's default numeric mode, N = 1's: the reference's sums (over the
    # ranks' tiles one after the other), exchanged through the ring (the ranks share the box's memory); the float64
    # step is reported beside it, as at N = 1
    assert d["parity_mode"].startswith("reference") and "callback" in d["config"]["exchange"] and "ring" in d["config"]["exchange"]
    assert d["shard_stats"]["ring_steps"] > 0 and d["shard_stats"]["collective_steps"] == 0
    # ... with every rank's inbox in its GPU's memory, mapped by the other process (HIP IPC; same-device here)
    assert d["shard_stats"]["rings_in_device_memory"] == 1 and "DEVICE memory" in d["config"]["exchange"]
    assert d["ms_per_step_reference_sums"] == d["ms_per_step"] and d["value_f64_tree"] > 0
    assert d["ms_per_step_min"] <= d["ms_per_step_median"] <= d["ms_per_step_max"]
    assert "cpu_baseline" not in d and "extra" not in d


@pytest.mark.gpu
@pytest.mark.timeout(300)
def test_bench_two_ranks_rehearsal():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, PCGX_BENCH_REHEARSE="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "8",
           "--warmup", "4", "--points", "100000"]
    _check(_run(cmd, env))


@pytest.mark.gpu
@pytest.mark.timeout(300)
def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` exactly as the driver starts the N = 1 run: the script launches the
    two ranks itself (torch.distributed.run as a child, before anything touches the GPU)."""
    env = dict(os.environ, PCGX_BENCH_REHEARSE="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "8", "--warmup", "4", "--points", "100000"]
    _check(_run(cmd, env))


@pytest.mark.gpu
@pytest.mark.timeout(300)
def test_bench_falls_back_when_rccl_refuses():
    """Two ranks on one device: RCCL's set-up fails (duplicate GPU) on the ranks; all of them must then
    agree on the host callback over gloo, finish, and say so in the line."""
    env = dict(os.environ, PCGX_BENCH_REHEARSE="2")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "8", "--warmup", "4", "--points", "100000"]
    d = _run(cmd, env)
    assert d["n_gpus"] == 2 and "RCCL set-up failed" in d["config"]["exchange"]
    assert d["final_value"] == d["final_value"] and d["final_value"] < 1e-3


@pytest.mark.gpu
@pytest.mark.timeout(300)
def test_bench_falls_back_when_rank_0_cannot_make_an_rccl_id():
    """No RCCL at all (PCGX_RCCL_DISABLE: pcgx_comm_unique_id fails on rank 0 before anybody has an id): the ranks
    agree on that BEFORE the others wait for the id, and take the host callback over gloo."""
    env = dict(os.environ, PCGX_BENCH_REHEARSE="2", PCGX_RCCL_DISABLE="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "8", "--warmup", "4", "--points", "100000"]
    d = _run(cmd, env)
    assert d["n_gpus"] == 2 and "RCCL set-up failed" in d["config"]["exchange"]
    assert d["final_value"] == d["final_value"] and d["final_value"] < 1e-3 and d["value_f64_tree_one_gpu_alone"] > 0


@pytest.mark.gpu
@pytest.mark.timeout(300)
def test_bench_takes_the_collectives_when_the_ring_breaks_in_the_warm_up():
    """The ring between the ranks has never run on two GPUs in this pipeline: should the warm-up break it on any rank (here:
    pretended, PCGX_BENCH_TEST_BREAK_RING), every rank goes on with the collective form of the same sums through a fresh
    communicator, and the line says so -- a line with the slower exchange, not no line."""
    env = dict(os.environ, PCGX_BENCH_REHEARSE="1", PCGX_BENCH_TEST_BREAK_RING="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "8", "--warmup", "4", "--points", "100000"]
    d = _run(cmd, env)
    assert d["n_gpus"] == 2 and "ring broke" in d["exchange_note"] and "all-reduces per step" in d["config"]["exchange"]
    assert d["shard_stats"]["collective_steps"] > 0 and d["shard_stats"]["ring_steps"] == 0
    assert d["final_value"] == d["final_value"] and d["final_value"] < 1e-3 and d["parity_mode"].startswith("reference")
