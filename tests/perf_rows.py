"""Per-row timings of SURVEY.md section 8 (hot path + the "next" rows built so far): the HIP path
through the C ABI next to the CPU oracle (C restatement of the reference, 1 thread) on a bounded
sample of the same workload.  Host-pointer entry points: the times INCLUDE the PCIe copies and the
host-side marshalling (the device-resident numbers of the headline path are bench.py's).
    python tests/perf_rows.py > profiles/rNN_rows.json
(lives under tests/ because it times and checks against the oracle, which only tests/, smoke() and
bench.py's cpu_baseline leg may load; not collected by pytest)"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import oracle as O
from pcgol_amd import _lib as L
from pcgol_amd import icp, kdtree, segmentation, synth, voxelgrid

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


# A2 MinMaxVec3 / A3-A4 VoxelGrid C1 (the reference's CPU-runnable config) and C3
c1 = synth.c1_voxel()
vg = voxelgrid.New(c1["leaf"])
g, out = timed(lambda: vg.Filter(c1["points"]))
t0 = time.perf_counter(); o = O.voxel_filter(c1["points"], len(c1["points"]), 12, 0, c1["leaf"]); cs = time.perf_counter() - t0
assert np.array_equal(out.Data, o)
row("A3/A4 VoxelGrid C1 100k leaf 0.05", "pc/filter/voxelgrid/voxelgrid.go:35-187", 1e5, "points", g, cs, 1e5)
c3 = synth.c3_voxel()
vg3 = voxelgrid.New(c3["leaf"])
# through the C ABI with the output buffer allocated (and touched) once, as a Go caller holds its slices: the
# Python mirror's fresh 120 MB result array costs more in page faults than the whole call (round 2 measured that)
import ctypes as C
_out3 = np.zeros_like(c3["points"])
_m3 = C.c_int64()
_leaf3 = (C.c_float * 3)(*c3["leaf"])
_chunk0 = (C.c_int32 * 3)(0, 0, 0)
def voxel_abi():
    L.check(L.lib().pcgx_voxel_filter(L.ptr(c3["points"]), len(c3["points"]), 12, 0, _leaf3, _chunk0, L.ptr(_out3), C.byref(_m3)))
    return _m3.value
g, m3 = timed(voxel_abi, reps=3)
out3 = vg3.Filter(c3["points"])
assert out3.Points == m3 and np.array_equal(out3.Data.view(np.float32).reshape(-1, 3), _out3[:m3])
sub = c3["points"][:1_000_000]
t0 = time.perf_counter(); O.voxel_filter(sub, len(sub), 12, 0, c3["leaf"]); cs = time.perf_counter() - t0
row("A3/A4 VoxelGrid C3 10M leaf 0.02 (host pointers)", "voxelgrid.go:35-187", 1e7, "points", g, cs, 1e6,
    "CPU sample: first 1M points (same cube)")
t0 = time.perf_counter(); O.minmax(c3["points"], len(c3["points"])); cs = time.perf_counter() - t0
mn, mx = np.empty(3, np.float32), np.empty(3, np.float32)
g, _ = timed(lambda: L.check(L.lib().pcgx_minmax(L.ptr(c3["points"]), len(c3["points"]), 12, 0, L.ptr(mn), L.ptr(mx))))
row("A2 MinMaxVec3 10M (host pointer)", "pc/minmax.go:9-26", 1e7, "points", g, cs, 1e7)
del c3, out3

# A5 kdtree.New, A6 Nearest (C2)
c2 = synth.c2_knn()
g, tree = timed(lambda: kdtree.New(c2["base"]), reps=3)
t0 = time.perf_counter(); otree = O.KDTree(c2["base"]); cs = time.perf_counter() - t0
row("A5 kdtree.New 1M (upload + device build + directory)", "pc/storage/kdtree/kdtree.go:33-56,348-370", 1e6, "points",
    g, cs, 1e6)
_ids = np.zeros(len(c2["queries"]), np.int64)
_dsq = np.zeros(len(c2["queries"]), np.float32)
def knn_abi():
    L.check(L.lib().pcgx_kdtree_nearest_batch(tree._h, L.ptr(c2["queries"]), len(c2["queries"]), c2["max_range"], 0.0,
                                              L.ptr(_ids), L.ptr(_dsq)))
    return _ids, _dsq
g, (ids, dsq) = timed(knn_abi)
nq = 100_000
t0 = time.perf_counter(); oi, od = otree.nearest_batch(c2["queries"][:nq], c2["max_range"]); cs = time.perf_counter() - t0
assert np.array_equal(ids[:nq], oi) and np.array_equal(dsq[:nq], od)
row("A6 KDTree.Nearest C2 1M x 1M (host pointers)", "kdtree.go:83-146,199-222", 1e6, "queries", g, cs, nq,
    "CPU sample: first 100k queries")

# N2 KDTree.Range
rq = c2["queries"][:200_000]
# through the C ABI with output buffers allocated once, as a Go caller would hold them (the Python
# mirror's fresh 33 MB result arrays cost more in page faults than the call takes)
offs, rid, rd = tree.RangeBatch(rq, 0.15)
_counts = np.zeros(len(rq), np.int64)
def range_abi():
    L.check(L.lib().pcgx_kdtree_range_count(tree._h, L.ptr(rq), len(rq), 0.15, L.ptr(_counts)))
    L.check(L.lib().pcgx_kdtree_range_fill(tree._h, L.ptr(rq), len(rq), 0.15, L.ptr(offs), L.ptr(rid), L.ptr(rd)))
g, _ = timed(range_abi, reps=5, warm=2)
nq = 20_000
t0 = time.perf_counter()
tot = 0
for q in rq[:nq]:
    tot += len(otree.range(q, 0.15)[0])
cs = time.perf_counter() - t0
assert tot == offs[nq]
row("N2 KDTree.Range 200k queries r=0.15 on 1M (%.1f neighbours/query)" % (offs[-1] / len(rq)), "kdtree.go:148-197",
    len(rq), "queries", g, cs, nq, "CPU sample: first 20k queries")

# N3 DeletePoint: delete 100k of 1M, then a 100k-query batch (host patching of the mirror tree + upload + walk)
gone = np.random.default_rng(1).permutation(1_000_000)[:100_000]
def del_and_query():
    t = kdtree.New(c2["base"])
    t.DeletePoints(gone)
    return t.NearestBatch(c2["queries"][:100_000], 1.0)
g_all, _ = timed(del_and_query, reps=2)
g_base, _ = timed(lambda: kdtree.New(c2["base"]).NearestBatch(c2["queries"][:100_000], 1.0), reps=2)
t0 = time.perf_counter()
for i in gone[:20_000]:
    otree.delete_point(int(i))
cs = time.perf_counter() - t0
row("N3 DeletePoint 100k of 1M (the reference's patching on the host mirror + upload of the patched tree)", "kdtree.go:224-332", 1e5, "deletions",
    max(g_all - g_base, 1e-6), cs, 2e4, "product time = (build + delete + query on the patched tree) - (build + query); CPU sample: 20k deletions")
del tree, otree

# N3 bucket grid + flood fill, N2 region growing
pts = synth.uniform_cloud(1_000_000, 10.0, 9)
size, origin, res = [128, 128, 128], [0, 0, 0], 0.08
def grid_all():
    v = segmentation.SegmentationVoxelGrid(res, size, origin)
    v.AddAll(pts)
    return v.Components()
g, comp = timed(grid_all, reps=2)
t0 = time.perf_counter()
og = O.BucketGrid(res, size, origin)
og.add_all(pts[:200_000])
og.segment(pts[0])
cs = time.perf_counter() - t0
row("N3 bucket grid Add x1M + flood-fill components of every voxel", "pc/storage/voxelgrid/voxelgrid.go:37-79; "
    "pc/segmentation/voxelgrid/voxelgrid.go:39-73", 1e6, "points", g, cs, 2e5,
    "CPU sample: Add of 200k points + ONE Segment(seed); the device labels every component")
rp = synth.uniform_cloud(300_000, 6.0, 10)
labels = np.random.default_rng(2).integers(0, 2, len(rp)).astype(np.uint32)
rt = kdtree.New(rp)
g, rcomp = timed(lambda: segmentation.RegionGrowing(rt, labels).Components(0.12), reps=2)
ort = O.KDTree(rp)
t0 = time.perf_counter(); seg = O.region_growing_segment(ort, labels, rp[0], 0.12); cs = time.perf_counter() - t0
assert np.array_equal(np.sort(seg), np.nonzero((rcomp == rcomp[0]) & (labels == labels[0]))[0]) or len(seg) == 0
row("N2 region growing: components of 300k points, maxRange 0.12", "pc/segmentation/regiongrowing/regiongrowing.go:23-56",
    len(rp), "points", g, cs, max(len(seg), 1), "CPU: ONE Segment(seed) reaching %d points; the device labels every region" % len(seg))

# N5 point-to-plane ICP (extension; no reference parity)
cp = synth.c4_plane(200_000)
pt = kdtree.New(cp["base"])
reg = icp.PointToPlaneICP(icp.PointToPlaneEvaluator(icp.NearestPointCorresponder(cp["max_dist"]), cp["normals"], 6),
                          icp.GaussNewtonUpdaterFactory(Threshold=cp["threshold"], MaxIteration=10))
g, (tr, st) = timed(lambda: reg.Fit(pt, cp["target"]), reps=2)
t0 = time.perf_counter()
o = O.plane_fit(O.KDTree(cp["base"]), cp["normals"], cp["target"], cp["max_dist"], 6, cp["threshold"], 0.0, 10)
cs = time.perf_counter() - t0
assert np.max(np.abs(tr - o["trans"])) <= 1e-5
row("N5 point-to-plane Fit 200k x 200k, 10 Gauss-Newton iterations (host pointers)", "extension: no reference code",
    2e6, "point-iterations", g, cs, 2e6, "oracle/plane_oracle.c (parity unpinned)")
