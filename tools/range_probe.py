"""Range batch timing probe: python tools/range_probe.py [nq] [radius]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pcgol_amd import synth, kdtree, _lib as L
nq = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
r = float(sys.argv[2]) if len(sys.argv) > 2 else 0.15
L.check(L.lib().pcgx_init(0))
c = synth.c2_knn()
t = kdtree.New(c["base"])
q = c["queries"][:nq]
for _ in range(2):
    offs, ids, dsq = t.RangeBatch(q, r)
t0 = time.perf_counter()
for _ in range(5):
    offs, ids, dsq = t.RangeBatch(q, r)
print("range batch nq=%d r=%g: %.3f ms, %.1f neighbours/query" % (nq, r, (time.perf_counter() - t0) / 5 * 1e3, offs[-1] / nq))
