import sys, os, ctypes
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, numpy as np
from pcgol_amd import synth, voxelgrid, _lib as L
n = 10_000_000
c3 = synth.c3_voxel(n)
L.check(L.lib().pcgx_init(0))
dp = torch.from_numpy(c3["points"]).cuda(); dout = torch.empty_like(dp)
vg = voxelgrid.New(c3["leaf"]); st = torch.cuda.current_stream().cuda_stream
raw = ctypes.CDLL(L.lib()._name)
NB = 13010
out = (ctypes.c_ulonglong * (NB * 10))()
for i in range(6):
    m = vg.FilterDev(dp.data_ptr(), n, 12, 0, dout.data_ptr(), st)
torch.cuda.synchronize()
raw.pcgx_debug_vbk(out, NB)
v = np.frombuffer(out, dtype=np.uint64).astype(np.float64).reshape(NB, 10) / 100.0
v = v[v[:, 7] > 0]
print("buckets with points:", len(v))
names = ["bounds", "points asked for, counts cleared, barrier", "ranks (ballots)", "cells scanned", "points to their places", "cell phase, stores issued", "barrier"]
prev = v[:, 0]
for k, nm in enumerate(names, 1):
    d = v[:, k] - prev
    print("%-44s mean %6.2f us  median %6.2f  p95 %6.2f" % (nm, d.mean(), np.median(d), np.percentile(d, 95)))
    prev = v[:, k]
life = v[:, 7] - v[:, 0]
print("a bucket: mean %.2f median %.2f us; kernel first in -> last out %.1f us; buckets in work at a time: %.0f" % (life.mean(), np.median(life), v[:, 7].max() - v[:, 0].min(), life.sum() / (v[:, 7].max() - v[:, 0].min())))
