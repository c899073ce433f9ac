"""rocprofv3 target: a few strict (set_strict 1) Fits on C4."""
import sys

sys.path.insert(0, ".")
from pcgol_amd import _lib as L, icp, kdtree, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 1
c = synth.c4_icp(n=n, width=10.0 * (n / 1e6) ** (1 / 3))
t = kdtree.New(c["base"])
s = icp.IcpSession(t, c["target"], c["max_dist"], c["min_pairs"], c["weight"], c["threshold"], c["max_iteration"])
s.set_strict(mode)
for rep in range(3):
    L.check(L.lib().pcgx_icp_session_reset(s._h, None))
    for k in range(c["max_iteration"]):
        s.step()
    L.check(L.lib().pcgx_sync(None))
print(s.result()[0])
