"""GPU: where DeletePoint's time goes -- the host patching, then the first query on the patched tree (upload + walk).
Usage: python tools/delete_breakdown.py"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from pcgol_amd import kdtree, synth  # noqa: E402

c2 = synth.c2_knn()
gone = np.random.default_rng(1).permutation(1_000_000)[:100_000]
for rep in range(2):
    t = kdtree.New(c2["base"])
    q = c2["queries"][:100_000]
    t.NearestBatch(q, 1.0)
    t0 = time.perf_counter()
    t.DeletePoints(gone[:1])
    t1 = time.perf_counter()
    t.DeletePoints(gone[1:])
    t2 = time.perf_counter()
    t.NearestBatch(q, 1.0)
    t3 = time.perf_counter()
    t.NearestBatch(q, 1.0)
    t4 = time.perf_counter()
    print("first deletion (builds the host mirror) %.1f ms, 99999 more %.1f ms (%.2f us each), first query %.1f ms, second %.1f ms" % (
        (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t2 - t1) * 1e6 / 99999, (t3 - t2) * 1e3, (t4 - t3) * 1e3))
