"""GPU: C4's iterations with the strict sums, a dump of the workgroups' stamps after each (PCGX_STRICT_TRACE=<file>,
set by the caller; no other clock reads: PCGX_STRICT_CLOCKS stays off).  Usage: python tools/strict_trace_run.py [n]"""
import sys

sys.path.insert(0, ".")
from pcgol_amd import icp, kdtree, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
c = synth.c4_icp(n=n, width=10.0 * (n / 1e6) ** (1 / 3))
t = kdtree.New(c["base"])
s = icp.IcpSession(t, c["target"], c["max_dist"], c["min_pairs"], c["weight"], c["threshold"], c["max_iteration"])
s.set_strict(1)
for k in range(c["max_iteration"]):
    s.step()
    s.strict_stats()
print("done")
