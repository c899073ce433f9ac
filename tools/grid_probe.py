"""Grid fast path of Nearest at C2 shape: timings with / without presort and PCGX_GRID=0, fallback count."""
import os, sys, time, ctypes as C
import numpy as np, torch
sys.path.insert(0, ".")
from pcgol_amd import kdtree, synth, _lib as L

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
w = float(sys.argv[2]) if len(sys.argv) > 2 else 10.0
mr = float(sys.argv[3]) if len(sys.argv) > 3 else 10.0
if os.environ.get("PROBE_SURFACE"):  # points on a curved surface, queries 2 cm off it
    pts = synth.surface_cloud(n, w, 2)[0]
    q = (synth.surface_cloud(n, w, 3)[0] + np.float32(0.02)).astype(np.float32)
else:
    pts = synth.uniform_cloud(n, w, 2)
    q = synth.uniform_cloud(n, w, 3)
t = kdtree.New(pts)
dq = torch.from_numpy(q).cuda()
ids = torch.empty(n, dtype=torch.int32, device="cuda")
dsq = torch.empty(n, dtype=torch.float32, device="cuda")
out = (C.c_int64 * 14)()
L.check(L.lib().pcgx_debug_grid_stats(t._h, L.ptr(dq.data_ptr()), n, mr, out))
print("grid stats: walk", out[0], "cells", out[1], "crowding", out[2] / 1000, "enabled", out[3], "why[1..7]", list(out)[5:12], "points/query", out[12] / n, "words/query", out[13] / n, flush=True)

def bench(label, presort, reps=10):
    for _ in range(2):
        t.NearestBatchDev(dq.data_ptr(), n, mr, ids.data_ptr(), dsq.data_ptr(), presort=presort)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        t.NearestBatchDev(dq.data_ptr(), n, mr, ids.data_ptr(), dsq.data_ptr(), presort=presort)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"{label}: {dt*1e3:.3f} ms ({n/dt/1e9:.3f} Gq/s)", flush=True)
    return ids.cpu().numpy().copy(), dsq.cpu().numpy().copy()

a = bench("grid presort", True)
if os.environ.get("PROBE_ONLY"):
    sys.exit(0)
b = bench("grid unsorted", False)
os.environ["PCGX_GRID"] = "0"
c = bench("walk presort", True)
d = bench("walk unsorted", False)
for x in (b, c, d):
    assert np.array_equal(a[0], x[0]) and np.array_equal(a[1], x[1])
print("all four agree")
