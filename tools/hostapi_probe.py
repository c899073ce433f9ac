import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pcgol_amd import synth, kdtree, voxelgrid, _lib as L
L.check(L.lib().pcgx_init(0))
c = synth.c2_knn()
t = kdtree.New(c["base"])
for nq in (1, 1000, 100_000, 1_000_000):
    q = c["queries"][:nq]
    for _ in range(2): t.NearestBatch(q, 10.0)
    t0 = time.perf_counter()
    for _ in range(5): t.NearestBatch(q, 10.0)
    print("nearest_batch nq=%d: %.3f ms" % (nq, (time.perf_counter() - t0) / 5 * 1e3))
c3 = synth.c3_voxel()
vg = voxelgrid.New(c3["leaf"])
for n in (1000, 1_000_000, 10_000_000):
    p = c3["points"][:n]
    for _ in range(2): vg.Filter(p)
    t0 = time.perf_counter()
    for _ in range(3): vg.Filter(p)
    print("voxel filter n=%d: %.3f ms" % (n, (time.perf_counter() - t0) / 3 * 1e3))
import torch
x = torch.empty(120_000_000, dtype=torch.uint8)
d = torch.empty(120_000_000, dtype=torch.uint8, device="cuda")
for _ in range(2): d.copy_(x)
torch.cuda.synchronize(); t0 = time.perf_counter(); d.copy_(x); torch.cuda.synchronize()
print("pageable H2D 120MB: %.2f ms" % ((time.perf_counter() - t0) * 1e3))
xp = x.pin_memory()
for _ in range(2): d.copy_(xp)
torch.cuda.synchronize(); t0 = time.perf_counter(); d.copy_(xp); torch.cuda.synchronize()
print("pinned H2D 120MB: %.2f ms" % ((time.perf_counter() - t0) * 1e3))
t0 = time.perf_counter(); xp.copy_(x); print("host memcpy 120MB: %.2f ms" % ((time.perf_counter() - t0) * 1e3))
