"""The N > 1 path of bench.py end to end on the one-GPU box: two ranks share cuda:0 and exchange
through gloo (PCGX_BENCH_REHEARSE=1; RCCL refuses two ranks on one device).  Checks what the
driver's multi-GPU run relies on: torchrun launch, tile construction, partials -> all-reduce ->
update on every rank, max-over-ranks timing, ONE JSON line from rank 0 with whole-job units."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


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
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=280, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1  # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 8 and d["warmup"] == 4 and d["scaling"] == "weak"
    assert d["config"]["target_points_per_gpu"] == 100000 and "x2" in d["config"]["parallelism"]
    assert abs(d["value"] - 2 * 100000 * 8 / (d["ms_per_step"] * 8 * 1e-3) / 1e6) < 1e-6 * d["value"]
    assert d["final_value"] == d["final_value"] and d["final_value"] < 1e-3  # the sharded Fit converges
    assert "cpu_baseline" not in d and "extra" not in d
