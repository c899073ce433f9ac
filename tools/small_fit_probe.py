"""The reference's ICP benchmark shape (icp_test.go:100-142) as a few session-resident Fits, for a profiler:
    python tools/small_fit_probe.py 4096 [fits]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pcgol_amd import icp, kdtree

f32 = np.float32
n_pts = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
fits = int(sys.argv[2]) if len(sys.argv) > 2 else 5
width = int(np.sqrt(float(n_pts)))
res = f32(10.0) / f32(width)
i = np.arange(n_pts)
bx = (res * (i // width).astype(f32) - f32(5)).astype(f32)
by = (res * (i % width).astype(f32) - f32(5)).astype(f32)
bz = np.where((bx > -1) & (bx < 1) & (by > -1) & (by < 1), f32(1), f32(0)).astype(f32)
base = np.ascontiguousarray(np.stack([bx, by, bz], axis=1))
target = (base + np.array([0.5, 0.3, -0.2], f32)).astype(f32)
thr = np.full(6, -1.0, f32)
t = kdtree.New(base, MinDistSq=float(res * res))
from pcgol_amd import _lib as L
reg = icp.PointToPointICPGradient(icp.PointToPointEvaluator(icp.NearestPointCorresponder(MaxDist=2.0), MinPairs=3),
                                  icp.GradientDescentUpdaterFactory(Threshold=thr, MaxIteration=10))
best = 1e9
for _ in range(fits):
    t0 = time.perf_counter()
    reg.Fit(t, target)
    best = min(best, time.perf_counter() - t0)
print("%d points: host-pointer 10-iteration Fit (session made, target uploaded, one launch, result read back) %.3f ms (best of %d)" % (n_pts, best * 1e3, fits))
if os.environ.get('PCGX_PROBE_HOST_ONLY'):
    sys.exit(0)
s = icp.IcpSession(t, target, 2.0, 3, None, thr, 10)
best = 1e9
for _ in range(fits):
    L.check(L.lib().pcgx_icp_session_reset(s._h, None))
    t0 = time.perf_counter()
    for _ in range(10):
        s.step()
    L.check(L.lib().pcgx_sync(None))
    best = min(best, time.perf_counter() - t0)
print("%d points: session, ten one-iteration launches %.3f ms (best of %d)" % (n_pts, best * 1e3, fits))
