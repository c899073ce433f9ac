"""Cost of the exact DeletePoint path: host patching (deleteNodeImpl mirror), upload, and Nearest /
Range on the explicit patched tree next to the implicit tree before deletion."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from pcgol_amd import kdtree, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
nq = n
pts = synth.uniform_cloud(n, 50.0, 1)
q = synth.uniform_cloud(nq, 50.0, 2)
t = kdtree.New(pts)
dq = torch.from_numpy(q).cuda()
ids = torch.empty(nq, dtype=torch.int32, device="cuda")
dsq = torch.empty(nq, dtype=torch.float32, device="cuda")

def bench(label, reps=5):
    t.NearestBatchDev(dq.data_ptr(), nq, 1.0, ids.data_ptr(), dsq.data_ptr(), presort=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        t.NearestBatchDev(dq.data_ptr(), nq, 1.0, ids.data_ptr(), dsq.data_ptr(), presort=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"{label}: nearest {dt*1e3:.3f} ms  ({nq/dt/1e9:.3f} Gq/s)", flush=True)

bench("implicit tree")
for frac in (0.001, 0.1):
    gone = np.random.default_rng(3).permutation(n)[: int(n * frac)]
    t0 = time.perf_counter()
    t.DeletePoints(gone)
    print(f"DeletePoints({len(gone)}) host patch: {(time.perf_counter()-t0)*1e3:.1f} ms", flush=True)
    t0 = time.perf_counter()
    t.NearestBatchDev(dq.data_ptr(), nq, 1.0, ids.data_ptr(), dsq.data_ptr(), presort=True)
    torch.cuda.synchronize()
    print(f"first query after deletion (upload): {(time.perf_counter()-t0)*1e3:.1f} ms", flush=True)
    bench(f"patched tree ({frac:g} deleted)")
    t0 = time.perf_counter()
    offs, rid, rd = t.RangeBatch(q[:100000], 0.5)
    print(f"RangeBatch 100k host call: {(time.perf_counter()-t0)*1e3:.1f} ms, {offs[-1]} hits", flush=True)
