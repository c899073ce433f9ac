"""GPU probe / rocprofv3 target: region growing components of 300k points (N2 row of tests/perf_rows.py)."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from pcgol_amd import kdtree, segmentation, synth  # noqa: E402

rp = synth.uniform_cloud(300_000, 6.0, 10)
labels = np.random.default_rng(2).integers(0, 2, len(rp)).astype(np.uint32)
rt = kdtree.New(rp)
for _ in range(3):
    c = segmentation.RegionGrowing(rt, labels).Components(0.12)
t0 = time.perf_counter()
for _ in range(3):
    c = segmentation.RegionGrowing(rt, labels).Components(0.12)
print("%.3f ms per call, %d components, checksum %d" % ((time.perf_counter() - t0) / 3 * 1e3, len(np.unique(c)), int(np.sum(c.astype(np.int64)))))
