"""One-shot host-pointer ICP calls (what the Go shim uses): python tools/fit_probe.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pcgol_amd import synth, kdtree, icp, _lib as L
L.check(L.lib().pcgx_init(0))
for n in (100_000, 1_000_000):
    c = synth.c4_icp(n=n, width=10.0 * (n / 1e6) ** (1 / 3))
    t = kdtree.New(c["base"])
    ev = icp.PointToPointEvaluator(icp.NearestPointCorresponder(MaxDist=c["max_dist"]), MinPairs=6)
    reg = icp.PointToPointICPGradient(ev, icp.GradientDescentUpdaterFactory(Weight=c["weight"], Threshold=c["threshold"],
                                                                           MaxIteration=20))
    for _ in range(2):
        reg.Fit(t, c["target"]); ev.Evaluate(t, c["target"])
    t0 = time.perf_counter()
    for _ in range(5): reg.Fit(t, c["target"])
    f = (time.perf_counter() - t0) / 5
    t0 = time.perf_counter()
    for _ in range(5): ev.Evaluate(t, c["target"])
    e = (time.perf_counter() - t0) / 5
    print("n=%d: Fit (20 iterations, host pointers) %.3f ms, Evaluate %.3f ms" % (n, f * 1e3, e * 1e3))
