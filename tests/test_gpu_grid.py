"""The certified grid fast path of KDTree.Nearest (csrc/knn_grid.h) against the oracle's walk.

The grid may only answer what does not depend on the reference's visit order; ties, DistSq ==
maxRange^2, sparse spots and non-finite queries must reach the tree walk.  Every case compares ID and
DistSq bit for bit with oracle/pcgol_oracle.c (kdtree.go:83-146) and with the walk-only library path
(PCGX_GRID=0)."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle as O
from pcgol_amd import _lib as L
from pcgol_amd import icp, kdtree, synth

pytestmark = pytest.mark.gpu
f32 = np.float32


def _grid_stats(t, q, max_range):
    import torch
    dq = torch.from_numpy(np.ascontiguousarray(q, f32)).cuda()
    out = (C.c_int64 * 14)()
    L.check(L.lib().pcgx_debug_grid_stats(t._h, L.ptr(dq.data_ptr()), len(q), max_range, out))
    return list(out)


def _check(pts, q, max_range, expect_grid=True):
    t, o = kdtree.New(pts), O.KDTree(pts)
    ids, dsq = t.NearestBatch(q, max_range)
    oi, od = o.nearest_batch(q, max_range)
    assert np.array_equal(dsq, od, equal_nan=True)
    assert np.array_equal(ids, oi)
    st = _grid_stats(t, q, max_range)
    assert st[3] == (1 if expect_grid else 0)
    return t, st


@pytest.mark.parametrize("n,width,seed", [(64, 1.0, 0), (1000, 3.0, 1), (40000, 10.0, 2), (200000, 25.0, 3)])
def test_uniform_clouds_queries_inside_and_outside(n, width, seed):
    pts = synth.uniform_cloud(n, width, 10 + seed)
    rng = np.random.default_rng(seed)
    q = np.concatenate([synth.uniform_cloud(20000, width, 20 + seed),
                        rng.uniform(-0.5 * width, 1.5 * width, (5000, 3)).astype(f32),   # around the box
                        rng.uniform(-50 * width, 50 * width, (200, 3)).astype(f32),      # far away
                        pts[:500]])                                                         # on base points
    for max_range in (width * 10, width * 0.05):
        t, st = _check(pts, q, max_range)
        assert st[0] < len(q)  # the grid answered something


def test_max_range_boundaries():
    """DistSq == maxRange^2 is accepted for a leaf and not for a pivot (kdtree.go:100-103 vs :117): the
    grid must hand those to the walk; just below / above it answers itself."""
    pts = synth.uniform_cloud(5000, 4.0, 5)
    o = O.KDTree(pts)
    q = synth.uniform_cloud(3000, 4.0, 6)
    _, d0 = o.nearest_batch(q, 100.0)
    t = kdtree.New(pts)
    for k in range(0, 3000, 7):
        r = f32(np.sqrt(np.float64(d0[k])))
        for mr in (r, np.nextafter(r, f32(0)), np.nextafter(r, f32(10))):
            got = t.Nearest(q[k], float(mr))
            oi, od = o.nearest_batch(q[k:k + 1], float(mr))
            assert got.ID == oi[0] and f32(got.DistSq) == od[0], (k, mr)


def test_lattice_every_query_tied_goes_to_the_walk():
    g = np.stack(np.meshgrid(*[np.arange(16)] * 3, indexing="ij"), -1).reshape(-1, 3).astype(f32)
    pts = g[np.random.default_rng(0).permutation(len(g))]
    q = (np.random.default_rng(1).integers(0, 31, (20000, 3)).astype(f32) * f32(0.5))
    t, st = _check(pts, q, 3.0)
    assert st[4 + 5] > 1000  # reason 5: ties


def test_duplicate_points_and_clusters():
    rng = np.random.default_rng(3)
    base = synth.uniform_cloud(3000, 5.0, 7)
    pts = np.concatenate([base, base[:1000], base[:300]])          # exact duplicates
    centres = rng.uniform(0, 5, (20, 3)).astype(f32)
    clusters = (centres[rng.integers(0, 20, 20000)] + rng.normal(0, 0.01, (20000, 3))).astype(f32)
    pts = np.concatenate([pts, clusters])[rng.permutation(24300)]
    q = np.concatenate([synth.uniform_cloud(10000, 5.0, 8), clusters[:5000] + f32(0.003)])
    t, o = kdtree.New(pts), O.KDTree(pts)
    ids, dsq = t.NearestBatch(q, 1.0)
    oi, od = o.nearest_batch(q, 1.0)
    assert np.array_equal(dsq, od) and np.array_equal(ids, oi)


@pytest.mark.parametrize("shape", ["plane", "line", "slab"])
def test_degenerate_extents(shape):
    rng = np.random.default_rng(4)
    pts = synth.uniform_cloud(20000, 8.0, 9)
    if shape == "plane":
        pts[:, 2] = f32(1.25)
    elif shape == "line":
        pts[:, 1] = f32(-3.0)
        pts[:, 2] = f32(0.5)
    else:
        pts[:, 0] = (pts[:, 0] * f32(0.001)).astype(f32)
    q = (pts[rng.integers(0, len(pts), 8000)] + rng.normal(0, 0.05, (8000, 3))).astype(f32)
    t, o = kdtree.New(pts), O.KDTree(pts)
    ids, dsq = t.NearestBatch(q, 2.0)
    oi, od = o.nearest_batch(q, 2.0)
    assert np.array_equal(dsq, od) and np.array_equal(ids, oi)


def test_surface_cloud_forced_grid_and_default(monkeypatch):
    """Surfaces in a box crowd the few cells they pass through: the build refines the cells until a
    point has ~3 neighbours in its own (so the grid stays in use); default, forced (PCGX_GRID=2) and
    walk-only runs all give the oracle's answers."""
    pts = np.ascontiguousarray(synth.surface_cloud(60000, 10.0, 11)[0], f32)
    q = (pts[::3] + f32(0.004)).astype(f32)
    o = O.KDTree(pts)
    oi, od = o.nearest_batch(q, 1.0)
    for mode in (None, "2", "0"):
        if mode is None:
            monkeypatch.delenv("PCGX_GRID", raising=False)
        else:
            monkeypatch.setenv("PCGX_GRID", mode)
        t = kdtree.New(pts)
        ids, dsq = t.NearestBatch(q, 1.0)
        assert np.array_equal(dsq, od) and np.array_equal(ids, oi), mode
        if mode is None:
            st = _grid_stats(t, q, 1.0)
            assert st[3] == 1 and st[2] <= 6000 and st[1] > 4 * len(pts)  # in use, crowding <= 6, refined cells


def test_non_finite_queries_reach_the_walk():
    pts = synth.uniform_cloud(5000, 2.0, 12)
    q = synth.uniform_cloud(200, 2.0, 13)
    q[3, 0] = np.nan
    q[7, 1] = np.inf
    q[11, 2] = -np.inf
    q[13] = np.nan
    _check(pts, q, 5.0)


def test_grid_and_walk_only_agree_at_c2_size(monkeypatch):
    pts = synth.uniform_cloud(1_000_000, 10.0, 2)
    q = synth.uniform_cloud(300_000, 10.0, 3)
    t = kdtree.New(pts)
    a = t.NearestBatch(q, 10.0)
    st = _grid_stats(t, q, 10.0)
    assert st[3] == 1 and st[0] < 100  # nearly everything certified
    monkeypatch.setenv("PCGX_GRID", "0")
    b = t.NearestBatch(q, 10.0)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    o = O.KDTree(pts)
    oi, od = o.nearest_batch(q[:20000], 10.0)
    assert np.array_equal(a[0][:20000], oi) and np.array_equal(a[1][:20000], od)


def test_icp_pairs_grid_vs_walk_vs_oracle(monkeypatch):
    """Iteration 0 (no hint) and a later iteration (hint = previous match) of a session: pairs from
    the grid pass equal the walk-only pairs and the oracle's."""
    c = synth.c4_icp(n=50000, width=3.7)
    t, o = kdtree.New(c["base"]), O.KDTree(c["base"])
    ob, ot, od = O.icp_pairs(o, c["target"], c["max_dist"])
    corr = icp.NearestPointCorresponder(MaxDist=c["max_dist"])
    b, ti, d = corr.PairsArrays(t, c["target"])
    assert np.array_equal(b, ob) and np.array_equal(ti, ot) and np.array_equal(d, od)

    def fit():
        s = icp.IcpSession(t, c["target"], c["max_dist"], c["min_pairs"], c["weight"], c["threshold"], 8,
                           SumsMode=icp.SumsF64Tree)
        for _ in range(8):
            s.step()
        r = s.result()
        sums = s.read_sums()
        s.close()
        return r, sums
    (tr1, st1, _), s1 = fit()
    monkeypatch.setenv("PCGX_GRID", "0")
    (tr0, st0, _), s0 = fit()
    # the float64 sums are added in a different (fixed) order: equal to ~1e-15, poses to a float32 ulp or two
    assert s1[9] == s0[9] and np.allclose(s1, s0, rtol=1e-12, atol=1e-12)
    assert np.max(np.abs(tr1 - tr0)) <= 1e-6


def test_random_clouds_grid_equals_walk_equals_oracle(monkeypatch):
    """Randomly drawn shapes -- anisotropic boxes, large offsets, mixtures of uniform / clustered /
    planar parts, queries near and far, small and large maxRange: the default path (grid where it is
    built) must equal the walk-only path bit for bit, and both the oracle on a sample."""
    rng = np.random.default_rng(2026)
    used = 0
    for case in range(40):
        n = int(rng.choice([300, 2000, 20000, 60000]))
        ext = rng.choice([0.05, 1.0, 7.0, 300.0], 3).astype(np.float64)
        off = rng.choice([0.0, -3.0, 1.0e3, -2.5e4], 3)
        parts = []
        kind = rng.integers(0, 4)
        u = rng.random((n, 3))
        if kind == 1:   # clusters
            c = rng.random((8, 3))
            u = c[rng.integers(0, 8, n)] + rng.normal(0, 0.01, (n, 3))
        elif kind == 2:  # half on a plane
            u[: n // 2, int(rng.integers(0, 3))] = 0.5
        elif kind == 3:  # density gradient
            u[:, 0] = u[:, 0] ** 3
        pts = (u * ext + off).astype(f32)
        nq = 6000
        q = np.concatenate([
            (pts[rng.integers(0, n, nq // 2)] + rng.normal(0, 0.02, (nq // 2, 3)) * ext).astype(f32),
            (rng.random((nq // 2, 3)) * 1.4 * ext - 0.2 * ext + off).astype(f32)])
        max_range = float(rng.choice([0.01, 0.2, 5.0]) * ext.max())
        monkeypatch.delenv("PCGX_GRID", raising=False)
        t = kdtree.New(pts)
        a = t.NearestBatch(q, max_range)
        used += _grid_stats(t, q, max_range)[3]
        monkeypatch.setenv("PCGX_GRID", "0")
        b = t.NearestBatch(q, max_range)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), case
        oi, od = O.KDTree(pts).nearest_batch(q[::6], max_range)
        assert np.array_equal(a[0][::6], oi) and np.array_equal(a[1][::6], od), case
    assert used >= 10  # the grid was in use for a good part of the cases


@pytest.mark.parametrize("max_range", [float("nan"), float("inf"), -0.3, 0.0, 1.0e20, 1.0e-30])
def test_odd_max_range_values(max_range):
    """maxRange only enters as maxRange * maxRange (kdtree.go:91): NaN compares false with everything,
    +inf / 1e20 square to +inf, 0 and 1e-30 leave nothing in range except exact hits."""
    pts = synth.uniform_cloud(4000, 2.0, 21)
    q = np.concatenate([synth.uniform_cloud(1500, 2.0, 22), pts[:200]])
    t, o = kdtree.New(pts), O.KDTree(pts)
    ids, dsq = t.NearestBatch(q, max_range)
    oi, od = o.nearest_batch(q, max_range)
    assert np.array_equal(ids, oi) and np.array_equal(dsq, od, equal_nan=True)


def test_icp_strict_fit_grid_equals_walk_on_random_setups(monkeypatch):
    """Whole strict Fits (sequential float32 sums: every pair of every iteration matters bit for bit)
    with and without the grid pass, on setups that exercise its branches: small and large MaxDist
    (many targets without a partner), partial overlap, outliers far from the base cloud, a lattice
    part (ties -> walk), few iterations and many."""
    rng = np.random.default_rng(77)
    for case in range(10):
        n = int(rng.choice([3000, 20000, 60000]))
        width = float(rng.choice([2.0, 6.0]))
        base = synth.uniform_cloud(n, width, 100 + case)
        if case % 3 == 2:  # a lattice corner: exact ties
            g = np.stack(np.meshgrid(*[np.arange(8)] * 3, indexing="ij"), -1).reshape(-1, 3).astype(f32) * f32(0.05)
            base = np.concatenate([base, g])
        pose = synth.icp_pose()
        tgt = synth.transform_points(pose, base[rng.permutation(len(base))[: n // 2]])
        extra = [tgt]
        if case % 2 == 0:  # outliers and a part that overlaps nothing
            extra.append((rng.random((500, 3)) * 3 * width - width).astype(f32))
            extra.append((rng.random((500, 3)) * 0.3 + width * 1.5).astype(f32))
        tgt = np.concatenate(extra).astype(f32)
        max_dist = float(rng.choice([0.03, 0.2, 1.0]))
        iters = int(rng.choice([3, 12]))
        w = np.full(6, 0.3, f32)
        th = np.full(6, -1.0, f32)

        def fit():
            t = kdtree.New(base)
            s = icp.IcpSession(t, tgt, max_dist, 6, w, th, iters)
            s.set_strict(True)
            try:
                for _ in range(iters):
                    s.step()
                tr, st, conv = s.result()
                out = (tr.copy(), st.NumIteration, float(st.Evaluated.Value), st.Evaluated.NumPairs)
            except icp.ErrNotEnoughPairs as e:
                out = ("not enough pairs", e.stat.NumIteration if hasattr(e, "stat") else None)
            s.close()
            return out
        monkeypatch.delenv("PCGX_GRID", raising=False)
        a = fit()
        monkeypatch.setenv("PCGX_GRID", "0")
        b = fit()
        if isinstance(a[0], str) or isinstance(b[0], str):
            assert a == b, case
        else:
            assert np.array_equal(a[0], b[0]) and a[1:] == b[1:], (case, a[1:], b[1:])


def test_large_batch_partitioned_by_coarse_cell_equals_the_oracle():
    """A batch large enough for the library to choose its own search order (>= 2^18 queries from host pointers): the
    queries are partitioned by an 8-bit coarse cell (csrc/knn_grid.hip, qp_* kernels) and searched in that order.
    Queries inside the cloud, far outside its box (they clamp to the faces' cells), on exact base points, NaN and
    infinite ones (the walk answers those), a maxRange that leaves most without a partner: ids and DistSq bits of the
    oracle, in the caller's order."""
    base = synth.uniform_cloud(200_000, 4.0, 31)
    rng = np.random.default_rng(5)
    n = 300_000
    q = synth.uniform_cloud(n, 4.0, 32)
    q[::7] = q[::7] * f32(3.0) - f32(4.0)                 # far outside on all sides
    q[1::1000] = base[rng.integers(0, len(base), len(q[1::1000]))]   # exact hits
    q[5::9973] = np.nan
    q[11::9973, 1] = np.inf
    q[13::9973, 2] = -np.inf
    t, o = kdtree.New(base), O.KDTree(base)
    # (three calls: the walk counts of consecutive calls take turns on two words that the calls leave at zero for one
    # another, Arena::zeroed_words -- the third call is back on the first one's)
    for max_range in (10.0, 0.02, 10.0):
        ids, dsq = t.NearestBatch(q, max_range)
        oi, od = o.nearest_batch(q, max_range)
        assert np.array_equal(ids, oi), max_range
        nan = np.isnan(od)
        assert np.array_equal(np.isnan(dsq), nan), max_range                       # (a NaN's payload is not pinned)
        assert np.array_equal(dsq[~nan].view(np.uint32), od[~nan].view(np.uint32)), max_range


def test_point_certificates_hold_against_brute_force(monkeypatch):
    """GridView::cert (csrc/knn_grid.hip, grid_cert_kernel; what lets the ICP loop keep a pair without a search): a query
    whose float32 DistSq to base point i is below cert[i] has i as its ONE nearest point -- every other point's
    float32 DistSq is strictly larger.  Checked against brute force on a cloud with a cluster, a sparse corner, twins
    (cert 0) and a lattice patch (many points at exactly equal distances), with queries placed just inside the bound."""
    from scipy.spatial import cKDTree
    rng = np.random.default_rng(11)
    pts = np.concatenate([
        rng.random((20000, 3)) * 4.0,                               # the cloud
        rng.random((3000, 3)) * 0.05 + 1.0,                         # a tight cluster
        rng.random((40, 3)) * 3.0 + 6.0,                            # a sparse corner
        np.stack(np.meshgrid(*[np.arange(8) * 0.125 + 2.0] * 3), -1).reshape(-1, 3),   # a lattice patch
    ]).astype(np.float32)
    pts = np.concatenate([pts, pts[:50]])                           # twins of the first fifty points
    monkeypatch.setenv("PCGX_GRID", "2")                            # (the grid is kept whatever the cluster does to its cells)
    t = kdtree.New(pts)
    cert = np.empty(len(pts), np.float32)
    L.check(L.lib().pcgx_debug_grid_cert(t._h, L.ptr(cert), len(pts)))
    assert np.all(cert >= 0.0) and np.all(np.isfinite(cert))
    assert np.all(cert[:50] == 0.0) and np.all(cert[-50:] == 0.0)   # twins: never certified
    tree = cKDTree(pts.astype(np.float64))
    d2nn = tree.query(pts.astype(np.float64), k=2)[0][:, 1] ** 2
    assert np.all(cert <= 0.25 * d2nn * (1.0 + 1e-5))               # never beyond half the way to the nearest other point
    assert np.mean(cert > 0.0) > 0.9                                 # and most points have one
    # queries at 0.97 of the bound, in random directions: float32 DistSq as the searches form it, brute force over all points
    idx = rng.choice(len(pts), 400, replace=False)
    idx = idx[cert[idx] > 0.0]
    dirs = rng.normal(size=(len(idx), 3))
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    q = (pts[idx].astype(np.float64) + dirs * (np.sqrt(cert[idx].astype(np.float64)) * 0.97)[:, None]).astype(np.float32)
    f = np.float32
    for k, i in enumerate(idx):
        d = pts - q[k]                                              # float32 throughout, (dx^2 + dy^2) + dz^2
        dsq = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
        if dsq[i] < cert[i]:
            others = np.delete(dsq, i)
            assert others.min() > dsq[i], (i, dsq[i], others.min(), cert[i])
    # and the library's own Nearest agrees on those queries
    ids, dsq = t.NearestBatch(q, 100.0)
    d = pts[idx] - q
    mine = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
    ok = mine < cert[idx]
    assert np.array_equal(np.asarray(ids)[ok], idx[ok]) and np.array_equal(np.asarray(dsq, np.float32)[ok], mine[ok])
