"""kNN walk-kernel timing probe: python tools/knn_probe.py [max_range] [presort 0/1] [n]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pcgol_amd import synth, kdtree, _lib as L
mr = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
presort = bool(int(sys.argv[2])) if len(sys.argv) > 2 else True
n = int(sys.argv[3]) if len(sys.argv) > 3 else 1_000_000
L.check(L.lib().pcgx_init(0))
c = synth.c2_knn(n, n)
t = kdtree.New(c["base"])
dq = torch.from_numpy(c["queries"]).cuda()
ids = torch.empty(n, dtype=torch.int32, device="cuda")
dsq = torch.empty(n, dtype=torch.float32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
L.prof_enable(True)
for _ in range(3):
    t.NearestBatchDev(dq.data_ptr(), n, mr, ids.data_ptr(), dsq.data_ptr(), presort, st)
torch.cuda.synchronize()
L.prof_reset()
t0 = time.perf_counter()
for _ in range(10):
    t.NearestBatchDev(dq.data_ptr(), n, mr, ids.data_ptr(), dsq.data_ptr(), presort, st)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 10
ms, k = L.prof_read(L.PROF_KNN_WALK)
print("max_range %g presort %d: call %.3f ms, walk kernel %.3f ms, matched %.1f%%" %
      (mr, presort, dt * 1e3, ms / max(k, 1), 100.0 * float((ids >= 0).float().mean())))
