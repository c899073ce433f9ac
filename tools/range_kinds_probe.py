"""GPU probe: RangeBatch on the grid against the walk (PCGX_RANGE_WALK=1) on three kinds of cloud."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from pcgol_amd import kdtree, synth  # noqa: E402

f32 = np.float32
rng = np.random.default_rng(77)
n, nq = 120_000, 30_000
cases = {
    "uniform": (synth.uniform_cloud(n, 10.0, 5), synth.uniform_cloud(nq, 10.0, 6), 0.25),
    "lattice": (rng.integers(0, 24, size=(n, 3)).astype(f32), rng.integers(0, 48, size=(nq, 3)).astype(f32) * f32(0.5), 1.6),
    "surface": (np.stack([rng.uniform(0, 10, n), rng.uniform(0, 10, n), rng.normal(0, 0.01, n)], axis=1).astype(f32),
                np.stack([rng.uniform(-1, 11, nq), rng.uniform(-1, 11, nq), rng.normal(0, 0.05, nq)], axis=1).astype(f32), 0.1),
}
for kind, (base, q, r) in cases.items():
    t = kdtree.New(base)
    out = {}
    for mode in ("grid", "walk"):
        if mode == "walk":
            os.environ["PCGX_RANGE_WALK"] = "1"
        else:
            os.environ.pop("PCGX_RANGE_WALK", None)
        for _ in range(3):  # (the library's arena settles on its size in the first two calls)
            t.RangeBatch(q, r)
        t0 = time.perf_counter()
        out[mode] = t.RangeBatch(q, r)
        dt = time.perf_counter() - t0
        print("%-8s %s: %8.2f ms, %d neighbours" % (kind, mode, dt * 1e3, len(out[mode][1])))
    same = all(np.array_equal(a, b) for a, b in zip(out["grid"], out["walk"]))
    d = out["grid"][2]
    o = out["grid"][0]
    ties = int(np.sum((d[1:] == d[:-1]) & (np.searchsorted(o, np.arange(1, len(d)), side="right") == np.searchsorted(o, np.arange(0, len(d) - 1), side="right"))))
    print("   identical: %s; adjacent equal DistSq inside a query: %d" % (same, ties))
