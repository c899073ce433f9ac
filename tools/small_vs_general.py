"""The one-launch Fit (csrc/icp_small.hip) against the general path on RANDOM clouds (synth.c4_icp: a surface in a box, the
caller's order random) and on the reference's benchmark shape, host-pointer Fits of 20 iterations: run once with
PCGX_ICP_SMALL=1 and once with =0 (the knob is read once a process):
    for s in 1 0; do PCGX_ICP_SMALL=$s python tools/small_vs_general.py; done"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pcgol_amd import icp, kdtree, synth

f32 = np.float32
shapes = [(1000, 1000), (2000, 2000), (4000, 4000), (8000, 8000), (16000, 16000), (32000, 16000), (32000, 2000), (1000, 16000), (32000, 8000)]
w, th = np.full(6, 0.3, f32), np.full(6, -1.0, f32)
for nb, nt in shapes:
    c = synth.c4_icp(n=max(nb, nt), width=2.0 + nb / 8000.0)
    base = np.ascontiguousarray(c["base"][:nb])
    target = np.ascontiguousarray(c["target"][:nt])
    t = kdtree.New(base)
    reg = icp.PointToPointICPGradient(icp.PointToPointEvaluator(icp.NearestPointCorresponder(MaxDist=0.5), MinPairs=6),
                                      icp.GradientDescentUpdaterFactory(Weight=w, Threshold=th, MaxIteration=20))
    best = 1e9
    for _ in range(6):
        t0 = time.perf_counter()
        reg.Fit(t, target)
        best = min(best, time.perf_counter() - t0)
    print("PCGX_ICP_SMALL=%s  base %6d  target %6d: 20-iteration host-pointer Fit %.3f ms" % (os.environ.get("PCGX_ICP_SMALL", "1"), nb, nt, best * 1e3), flush=True)
