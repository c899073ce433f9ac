"""rocprofv3 target: RangeBatch 200k queries, r = 0.15, on the 1M cloud (count + fill)."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from pcgol_amd import _lib as L, kdtree, synth  # noqa: E402

base = synth.uniform_cloud(1_000_000, 10.0, 2)
q = synth.uniform_cloud(200_000, 10.0, 3)
t = kdtree.New(base)
for rep in range(4):
    t0 = time.perf_counter()
    r = t.RangeBatch(q, 0.15)
    dt = time.perf_counter() - t0
print("RangeBatch 200k: %.3f ms host-pointer call, mean neighbours %.2f" % (dt * 1e3, np.mean([len(x) for x in r[:2000]])))
