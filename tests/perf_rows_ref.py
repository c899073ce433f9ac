"""Rows for the reference's OWN benchmarks (BenchmarkKDTree_Nearest, kdtree_test.go:1040-1086;
BenchmarkPointToPointICPGradient, icp_test.go:100-142), shape for shape, next to the CPU oracle (C restatement of the
reference, 1 thread) on the same inputs; every answer checked against the oracle.  Same line format as perf_rows.py:
    python tests/perf_rows_ref.py >> profiles/rNN_rows.json
(under tests/ because it loads the oracle; not collected by pytest)"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import oracle as O
from pcgol_amd import _lib as L
from pcgol_amd import icp, kdtree, synth

L.check(L.lib().pcgx_init(0))
rows = []


def timed(fn, reps=3, warm=1):
    for _ in range(warm):
        fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = fn()
    return (time.perf_counter() - t0) / reps, r


def row(name, ref, unit_count, unit, gpu_s, cpu_s, cpu_units, note=""):
    rows.append({"row": name, "reference": ref, "unit": unit, "gpu_units_per_s": unit_count / gpu_s,
                 "gpu_ms": gpu_s * 1e3, "cpu_oracle_units_per_s": cpu_units / cpu_s, "cpu_sample_units": cpu_units,
                 "speedup": (unit_count / gpu_s) / (cpu_units / cpu_s), "note": note})
    print(json.dumps(rows[-1]), flush=True)

# ---- the reference's OWN benchmarks, shape for shape ------------------------------------------------
# BenchmarkKDTree_Nearest (pc/storage/kdtree/kdtree_test.go:1040-1086): width 10, 100 targets, maxRange = width,
# N in {100, 1k, 10k, 100k}, MinDistSq in {0, 0.1, 0.01} (> 0: the approximate search, answered wholly by the walk in
# the reference's visit order).  The reference times ONE Nearest per op; here: one blocking single-point call per op (what
# a caller that loops over storage.Search.Nearest gets: launch + two PCIe round trips), the 100 targets as one batch, and
# a 1M-query batch (the seam the hot callers use), next to the oracle's loop.  Every answer is checked against the oracle.
for min_dist_sq in (0.0, 0.1, 0.01):
    for n_pts in (100, 1000, 10_000, 100_000):
        cloud = synth.uniform_cloud(n_pts, 10.0, 60 + n_pts % 7)
        targets = synth.uniform_cloud(100, 10.0, 77)
        big = synth.uniform_cloud(1_000_000, 10.0, 78)
        gt = kdtree.New(cloud, MinDistSq=min_dist_sq)
        ot = O.KDTree(cloud, min_dist_sq)
        g1, _ = timed(lambda: [gt.Nearest(p, 10.0) for p in targets[:20]], reps=2)
        g100, (bi, bd) = timed(lambda: gt.NearestBatch(targets, 10.0), reps=5)
        gbig, (gi, gd) = timed(lambda: gt.NearestBatch(big, 10.0), reps=2)
        t0 = time.perf_counter()
        reps = 200 if n_pts <= 10_000 else 50
        for _ in range(reps):
            oi, od = ot.nearest_batch(targets, 10.0)
        cs = (time.perf_counter() - t0) / reps
        assert np.array_equal(bi, oi) and np.array_equal(bd, od)
        so, sd = ot.nearest_batch(big[:20_000], 10.0)
        assert np.array_equal(gi[:20_000], so) and np.array_equal(gd[:20_000], sd)
        row("BenchmarkKDTree_Nearest minDistSq=%.2f %dpoints: 1M-query batch" % (min_dist_sq, n_pts),
            "pc/storage/kdtree/kdtree_test.go:1040-1086", 1e6, "queries", gbig, cs, 100,
            "ns per Nearest -- oracle loop %.0f; GPU: one blocking single-point call %.0f, the 100 targets as one batch %.0f, "
            "in the 1M batch %.1f (host pointers, PCIe included)" % (cs / 100 * 1e9, g1 / 20 * 1e9, g100 / 100 * 1e9, gbig / 1e6 * 1e9))
        del gt, ot

# BenchmarkPointToPointICPGradient (pc/registration/icp/icp_test.go:100-142): the 10 x 10 m ground grid with a 2 x 2 x 1 box,
# target = base + (0.5, 0.3, -0.2), MaxDist 2, MinPairs 3, Threshold -1, MaxIteration 10, MinDistSq = res^2; one Fit per op.
for n_pts in (1024, 4096, 16384):
    width = int(np.sqrt(float(n_pts)))
    res = np.float32(10.0) / np.float32(width)
    i = np.arange(n_pts)
    bx = (res * (i // width).astype(np.float32) - np.float32(5)).astype(np.float32)
    by = (res * (i % width).astype(np.float32) - np.float32(5)).astype(np.float32)
    bz = np.where((bx > -1) & (bx < 1) & (by > -1) & (by < 1), np.float32(1), np.float32(0)).astype(np.float32)
    gbase = np.ascontiguousarray(np.stack([bx, by, bz], axis=1))
    gtarget = (gbase + np.array([0.5, 0.3, -0.2], np.float32)).astype(np.float32)
    mds = float(res * res)
    gt = kdtree.New(gbase, MinDistSq=mds)
    ot = O.KDTree(gbase, mds)
    thr = np.full(6, -1.0, np.float32)
    reg = icp.PointToPointICPGradient(icp.PointToPointEvaluator(icp.NearestPointCorresponder(MaxDist=2.0), MinPairs=3),
                                      icp.GradientDescentUpdaterFactory(Threshold=thr, MaxIteration=10))
    g, (tr, st) = timed(lambda: reg.Fit(gt, gtarget), reps=5)
    t0 = time.perf_counter()
    for _ in range(3):
        o = O.icp_fit(ot, gtarget, 2.0, 3, None, thr, 10, sums_mode=0)
    cs = (time.perf_counter() - t0) / 3
    assert st.NumIteration == o["num_iteration"] == 10
    assert np.array_equal(np.asarray(tr).ravel(), np.asarray(o["trans"]).ravel()) and st.Evaluated.Value == o["value"]
    # the loop resident on the device (a session: targets uploaded once), per Fit
    s = icp.IcpSession(gt, gtarget, 2.0, 3, None, thr, 10)
    def resident_fit():
        L.check(L.lib().pcgx_icp_session_reset(s._h, None))
        for _ in range(10):
            s.step()
        L.check(L.lib().pcgx_sync(None))
    gs, _ = timed(resident_fit, reps=10, warm=2)
    s.close()
    row("BenchmarkPointToPointICPGradient Points%d: one 10-iteration Fit" % n_pts, "pc/registration/icp/icp_test.go:100-142",
        1, "fits", g, cs, 1,
        "ms per Fit -- oracle %.3f; GPU host-pointer Fit %.3f (session made, target uploaded, result read back per call), "
        "a session's ten one-iteration launches %.3f; MinDistSq = res^2: the approximate search, every pair the reference-order "
        "walk's (picked out of all distances by visit order, csrc/icp_small.hip); "
        "pose and Value bit-identical to the oracle" % (cs * 1e3, g * 1e3, gs * 1e3))
    del gt, ot
