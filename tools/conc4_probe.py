"""Four host-pointer Fits from four host threads (bench.py's extra.icp_c4_concurrent4) on their own: for a kernel trace."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pcgol_amd import synth, icp, kdtree
c4 = synth.c4_icp()
reg = icp.PointToPointICPGradient(
    icp.PointToPointEvaluator(icp.NearestPointCorresponder(MaxDist=c4["max_dist"]), MinPairs=c4["min_pairs"]),
    icp.GradientDescentUpdaterFactory(Weight=c4["weight"], Threshold=c4["threshold"], MaxIteration=c4["max_iteration"]))
tree = kdtree.New(c4["base"])
reg.Fit(tree, c4["target"])
t0 = time.perf_counter(); reg.Fit(tree, c4["target"]); one = time.perf_counter() - t0
nthreads = int(sys.argv[1]) if len(sys.argv) > 1 else 4
reps = []
for rep in range(8):
    th = [threading.Thread(target=lambda: reg.Fit(tree, c4["target"])) for _ in range(nthreads)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    four = time.perf_counter() - t0
    reps.append(four * 1e3)
print("one fit %.3f ms; %d at once, round by round: %s ms; last %.3f (%.2f x one), best %.3f (%.2f x one)" % (
    one * 1e3, nthreads, " ".join("%.2f" % r for r in reps), four * 1e3, four / one, min(reps), min(reps) / (one * 1e3)))
