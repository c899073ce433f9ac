"""Per-workgroup wall clock stamps of an instrumented kernel (csrc/wg_stamps.h): builds nothing itself -- run it with a
library built with -DPCGX_STAMPS:
    bash tools/mk_variant.sh stamps "-DPCGX_STAMPS" && PCGX_LIB=experiments/ab/libpcgx_stamps.so python tools/stamps.py vb_bucket
Runs the C3 filter a few times, reads the last call's stamps, prints every phase's time per workgroup (mean / median /
p95), the workgroups' lives and how many were at work at a time."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pcgol_amd import synth, voxelgrid, _lib as L

KERNELS = {  # name -> (stamps per workgroup, rows to read, phase names (stamp k - stamp k-1), driver)
    "vb_bucket": (8, 65536, ["bounds", "points asked for, counts cleared, barrier", "ranks (ballots)", "cells scanned",
                             "points to their places", "cell phase, stores issued"], "voxel"),
    "small_fit": (8, 256, ["pose in, query re-projected", "chunks (distances, minima)", "minima met; first workgroup of a group: decides, terms out",
                           "(first wave: nothing) sums are other waves'", "workgroup 0: sums in, update, pose out"], "small"),
}


def run_small():
    """the reference's benchmark shape (icp_test.go:100-142), PCGX_STAMPS_POINTS points (1024)"""
    from pcgol_amd import icp, kdtree
    n_pts = int(os.environ.get("PCGX_STAMPS_POINTS", "1024"))
    width = int(np.sqrt(float(n_pts)))
    f32 = np.float32
    res = f32(10.0) / f32(width)
    i = np.arange(n_pts)
    bx = (res * (i // width).astype(f32) - f32(5)).astype(f32)
    by = (res * (i % width).astype(f32) - f32(5)).astype(f32)
    bz = np.where((bx > -1) & (bx < 1) & (by > -1) & (by < 1), f32(1), f32(0)).astype(f32)
    base = np.ascontiguousarray(np.stack([bx, by, bz], axis=1))
    target = (base + np.array([0.5, 0.3, -0.2], f32)).astype(f32)
    reg = icp.PointToPointICPGradient(icp.PointToPointEvaluator(icp.NearestPointCorresponder(MaxDist=2.0), MinPairs=3),
                                      icp.GradientDescentUpdaterFactory(Threshold=np.full(6, -1.0, f32), MaxIteration=10))
    t = kdtree.New(base, MinDistSq=float(res * res))
    for _ in range(3):
        reg.Fit(t, target)
    raw = ctypes.CDLL(L.lib()._name)
    cnt = (ctypes.c_ulonglong * 8)()
    if getattr(raw, "pcgx_debug_small_counts", None) is not None:
        raw.pcgx_debug_small_counts(cnt, 1)
        reg.Fit(t, target)
        raw.pcgx_debug_small_counts(cnt, 0)
        print("  one Fit (10 iterations, %d groups of 64 targets): %d chunks looked at, %d ruled out whole, %d gone through" % ((n_pts + 63) // 64, cnt[0], cnt[1], cnt[2]))
        fl = (ctypes.c_ulonglong * 16)()
        if getattr(raw, "pcgx_debug_small_fails", None) is not None and raw.pcgx_debug_small_fails(fl) == 0:
            print("  the stamped iteration, tiles added term by term because their record did not cover the state, row by row (of them without a window): "
                  + " ".join("%d(%d)" % (fl[r] >> 8, fl[r] & 255) for r in range(9)))
        it = (ctypes.c_ulonglong * 64)()
        if getattr(raw, "pcgx_debug_small_iter_times", None) is not None and raw.pcgx_debug_small_iter_times(it) == 0:
            tt = [x / 100.0 for x in it]
            print("  the last Fit's launch, workgroup 0's first wave: in the loop %.2f us after its start; iterations %s us; out of the loop %.2f us after the last one's top, the state mailed %.2f us later"
                  % (tt[1] - tt[0], " ".join("%.1f" % (tt[k + 1] - tt[k]) for k in range(1, 10)), tt[61] - tt[10], tt[62] - tt[61]))
    else:
        reg.Fit(t, target)


def run_voxel():
    n = 10_000_000
    c3 = synth.c3_voxel(n)
    dp = torch.from_numpy(c3["points"]).cuda()
    dout = torch.empty_like(dp)
    vg = voxelgrid.New(c3["leaf"])
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(6):
        vg.FilterDev(dp.data_ptr(), n, 12, 0, dout.data_ptr(), st)
    torch.cuda.synchronize()


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "vb_bucket"
    per, rows, phases, driver = KERNELS[name]
    L.check(L.lib().pcgx_init(0))
    raw = ctypes.CDLL(L.lib()._name)
    fn = getattr(raw, "pcgx_debug_stamps_" + name, None)
    if fn is None:
        raise SystemExit("this library has no stamps for %s: build it with -DPCGX_STAMPS (tools/mk_variant.sh) and set PCGX_LIB" % name)
    {"voxel": run_voxel, "small": run_small}[driver]()
    out = (ctypes.c_ulonglong * (rows * per))()
    if fn(out, rows) != 0:
        raise SystemExit("reading the stamps failed")
    v = np.frombuffer(out, dtype=np.uint64).astype(np.float64).reshape(rows, per) / 100.0  # us
    last = len(phases)
    v = v[(v[:, 0] > 0) & (v[:, last] > 0)]
    print("%s: %d workgroup-units stamped" % (name, len(v)))
    for k, nm in enumerate(phases, 1):
        d = v[:, k] - v[:, k - 1]
        print("  %-44s mean %6.2f us  median %6.2f  p95 %6.2f" % (nm, d.mean(), np.median(d), np.percentile(d, 95)))
    if name == "small_fit":  # workgroup 0's first wave is the iteration's critical path
        w0 = np.frombuffer(out, dtype=np.uint64).astype(np.float64).reshape(rows, per)[0] / 100.0
        print("  workgroup 0: " + "  ".join("%.2f" % (w0[k] - w0[k - 1]) for k in range(1, last + 1)) + "  = %.2f us" % (w0[last] - w0[0]))
        allw = np.frombuffer(out, dtype=np.uint64).astype(np.float64).reshape(rows, per) / 100.0
        t0 = w0[0]
        print("  since workgroup 0 had the pose: its terms out %.2f; row 0's chain: begins to look for terms %.2f, first batch in %.2f, sum out %.2f; "
              "updater: sums in %.2f, pose out %.2f; a wave of workgroup 3 (flat): %d chunks, %.0f shader cycles a chunk gone through, %.0f waiting for its records, %.0f all in all" % (w0[3] - t0, allw[1][6] - t0, allw[1][7] - t0, allw[2][6] - t0, allw[2][7] - t0, w0[5] - t0, allw[4][6] * 100, allw[3][6] * 100 / max(allw[4][6] * 100, 1), allw[3][7] * 100 / max(allw[4][6] * 100, 1), allw[4][7] * 100 / max(allw[4][6] * 100, 1)))
    life = v[:, last] - v[:, 0]
    span = v[:, last].max() - v[:, 0].min()
    print("  a unit: mean %.2f median %.2f us; first in -> last out %.1f us; at work at a time: %.0f" % (life.mean(), np.median(life), span, life.sum() / span))


if __name__ == "__main__":
    main()
