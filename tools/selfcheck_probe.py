"""GPU probe: the chain kernel's in-kernel self-check (PCGX_STRICT_SELFCHECK: every step of the walk re-derived term by
term) on a target of several chunks; prints the mismatch counters and how many walkers gave up their wait.
    python tools/selfcheck_probe.py [n_targets]"""
import os
import sys

import numpy as np

os.environ["PCGX_STRICT_SELFCHECK"] = "1"
sys.path.insert(0, ".")
from pcgol_amd import icp, kdtree, synth  # noqa: E402

n_t = int(sys.argv[1]) if len(sys.argv) > 1 else 2_600_000
n_base = 300_000
base = synth.uniform_cloud(n_base, 6.7, 51)
rng = np.random.default_rng(53)
target = synth.transform_points(synth.icp_pose(), base[rng.integers(0, n_base, n_t)] +
                                rng.uniform(-0.01, 0.01, (n_t, 3)).astype(np.float32)).astype(np.float32)
t = kdtree.New(base)
cfg = dict(MaxDist=0.5, MinPairs=6, Weight=np.full(6, 0.3, np.float32), Threshold=np.full(6, -1.0, np.float32), MaxIteration=20)
a, b = icp.IcpSession(t, target, **cfg), icp.IcpSession(t, target, **cfg)
a.set_strict(1)
b.set_strict(2)
for k in range(4):
    a.step()
    b.step()
    st = a.strict_stats()
    print(k, "equal", np.array_equal(a.read_sums().view(np.uint64), b.read_sums().view(np.uint64)), "selfcheck", st[12:16], "resolved", st[2], "no aux", st[5],
          "walked alone", st[62], "gave up exchange", st[63], flush=True)
