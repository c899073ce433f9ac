"""How well the one-query-per-lane grid scans fill their waves: useful point records vs the
lane-slots the scan loops run (a wave runs its loop as long as its busiest lane needs)."""
import ctypes as C
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from pcgol_amd import _lib as L, icp, kdtree, synth  # noqa: E402

n = 1_000_000
pts = synth.uniform_cloud(n, 10.0, 2)
q = synth.uniform_cloud(n, 10.0, 3)
t = kdtree.New(pts)
# queries in cell order, as the presorted product path hands them to the kernel
cell = np.floor(q / np.float32(10.0 / 80)).astype(np.int64)
order = np.lexsort((cell[:, 0], cell[:, 1], cell[:, 2]))
for label, qq in (("kNN C2, queries in caller order", q), ("kNN C2, queries in cell order", q[order])):
    dq = torch.from_numpy(np.ascontiguousarray(qq)).cuda()
    out = (C.c_int64 * 14)()
    L.check(L.lib().pcgx_debug_grid_stats(t._h, L.ptr(dq.data_ptr()), n, 10.0, out))
    print(f"{label}: records/query {out[12] / n:.2f}, lane-slots/query {out[4] / n:.2f}, fill {out[12] / max(out[4], 1):.3f}")
c = synth.c4_icp()
t = kdtree.New(c["base"])
s = icp.IcpSession(t, c["target"], c["max_dist"], c["min_pairs"], c["weight"], c["threshold"], c["max_iteration"])
for k in range(6):
    g = s.grid_stats()
    print(f"ICP C4 iteration {k}: records/target {g[2] / g[0]:.2f}, lane-slots/target {g[4] / g[0]:.2f}, fill {g[2] / max(g[4], 1):.3f}")
    s.step()
