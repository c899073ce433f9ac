"""Host-pointer seams: ring staging (default) against direct copies (PCGX_STAGE=0), on a buffer the runtime has
seen before and on fresh ones (what a Go caller's slices usually are).  python tools/stage_probe.py"""
import ctypes as C, os, sys, time
sys.path.insert(0, ".")
import numpy as np
from pcgol_amd import _lib as L, synth, kdtree
L.check(L.lib().pcgx_init(0))
c3 = synth.c3_voxel()
pts = c3["points"]
leaf = (C.c_float * 3)(*c3["leaf"]); chunk = (C.c_int32 * 3)(0, 0, 0)
def filt(src, out):
    m = C.c_int64()
    L.check(L.lib().pcgx_voxel_filter(L.ptr(src), len(src), 12, 0, leaf, chunk, L.ptr(out), C.byref(m)))
    return m.value
out = np.zeros_like(pts)
filt(pts, out)
for name, fresh_in, fresh_out in (("same buffers", False, False), ("fresh input", True, False), ("fresh input and output (touched)", True, True)):
    ts = []
    for rep in range(4):
        src = pts.copy() if fresh_in else pts
        dst = np.zeros_like(pts) if fresh_out else out
        t0 = time.perf_counter(); filt(src, dst); ts.append(time.perf_counter() - t0)
    print("voxel C3 host pointers, %s: %s ms" % (name, " ".join("%.2f" % (t * 1e3) for t in ts)))
c2 = synth.c2_knn()
t = kdtree.New(c2["base"])
ids = np.zeros(len(c2["queries"]), np.int64); dsq = np.zeros(len(c2["queries"]), np.float32)
def knn(q):
    L.check(L.lib().pcgx_kdtree_nearest_batch(t._h, L.ptr(q), len(q), 10.0, 0.0, L.ptr(ids), L.ptr(dsq)))
knn(c2["queries"])
for name, fresh in (("same buffers", False), ("fresh queries", True)):
    ts = []
    for rep in range(4):
        q = c2["queries"].copy() if fresh else c2["queries"]
        t0 = time.perf_counter(); knn(q); ts.append(time.perf_counter() - t0)
    print("kNN C2 host pointers, %s: %s ms" % (name, " ".join("%.2f" % (x * 1e3) for x in ts)))
