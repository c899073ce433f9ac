"""kNN C2 presorted call: per-kernel device times (library profiling scopes)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from pcgol_amd import kdtree, synth, _lib as L
n = 1_000_000
pts = synth.uniform_cloud(n, 10.0, 2); q = synth.uniform_cloud(n, 10.0, 3)
t = kdtree.New(pts)
dq = torch.from_numpy(q).cuda()
ids = torch.empty(n, dtype=torch.int32, device="cuda"); dsq = torch.empty(n, dtype=torch.float32, device="cuda")
for _ in range(3):
    t.NearestBatchDev(dq.data_ptr(), n, 10.0, ids.data_ptr(), dsq.data_ptr(), presort=True)
torch.cuda.synchronize()
L.prof_enable(1); L.prof_reset()
t0 = time.perf_counter()
for _ in range(10):
    t.NearestBatchDev(dq.data_ptr(), n, 10.0, ids.data_ptr(), dsq.data_ptr(), presort=True)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 10
ms, cnt = L.prof_read(L.PROF_KNN_GRID)
print(f"call {dt*1e3:.3f} ms (with timing events), grid kernel {ms/max(cnt,1)*1e3:.1f} us ({cnt} launches)")
L.prof_enable(0)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    t.NearestBatchDev(dq.data_ptr(), n, 10.0, ids.data_ptr(), dsq.data_ptr(), presort=True)
torch.cuda.synchronize()
print(f"call {(time.perf_counter() - t0) / 20 * 1e3:.4f} ms (profiling off)")
