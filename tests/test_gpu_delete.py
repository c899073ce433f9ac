"""GPU parity of KDTree.DeletePoint (kdtree.go:322-332) through the C ABI.

The oracle restates the reference's deleteNodeImpl / findMinimumImpl (pinned by the reference's
exact trees after deletion, tests/test_oracle_golden.py).  The product keeps the same patched tree
(csrc/knn_explicit.hip): its dump must equal the reference's expected trees node for node, and
Nearest / Range on it return ID and DistSq exactly as the oracle's patched tree does -- on lattice
clouds full of exact-distance ties and with MinDistSq > 0 too, where the answer depends on the
tree's shape."""
import numpy as np
import pytest

import oracle as O
from pcgol_amd import icp, kdtree, synth

pytestmark = pytest.mark.gpu
f32 = np.float32


def test_delete_tables_nearest(golden):
    """After every step of every known-answer deletion sequence (kdtree_test.go:413-729) the device
    answers all Nearest / Range queries like the oracle's patched tree."""
    g = golden("ref_kdtree.json")
    pts = np.array(g["test_cloud"]["points"], f32)
    queries = np.array([c["p"] for c in g["nearest"]["cases"]] + pts.tolist(), f32) + f32(0.013)
    for name, steps in g["delete_point"]["sequences"].items():
        t, o = kdtree.New(pts), O.KDTree(pts)
        for st in steps:
            if st["has_error"]:
                with pytest.raises(IndexError):
                    t.DeletePoint(st["pid"])
                with pytest.raises(IndexError):
                    o.delete_point(st["pid"])
            else:
                t.DeletePoint(st["pid"])
                o.delete_point(st["pid"])
            ids, dsq = t.NearestBatch(queries, 10.0)
            oi, od = o.nearest_batch(queries, 10.0)
            assert np.array_equal(ids, oi) and np.array_equal(dsq, od), (name, st["pid"])
            assert t.Len() == len(pts)  # the accessor is unchanged
            for q in queries[:4]:
                r = t.Range(q, 2.5)
                ri, rd = o.range(q, 2.5)
                assert sorted((n.ID, float(n.DistSq)) for n in r) == sorted(zip(ri.tolist(), rd.tolist()))


def test_delete_tables_tree_equals_the_reference_trees(golden):
    """kdtree_test.go:413-729: after every DeletePoint the PRODUCT's tree is the expected tree."""
    g = golden("ref_kdtree.json")
    pts = np.array(g["test_cloud"]["points"], f32)
    assert kdtree.New(pts).Tree() == g["expected_tree"]["root"]
    for name, steps in g["delete_point"]["sequences"].items():
        t, o = kdtree.New(pts), O.KDTree(pts)
        for st in steps:
            try:
                t.DeletePoint(st["pid"])
                assert not st["has_error"], name
                o.delete_point(st["pid"])
            except IndexError:
                assert st["has_error"], name
            assert t.Tree() == st["tree"], (name, st["pid"])
            assert t.MaxDepth() == o.max_depth()


def _lattice(side, seed):
    g = np.stack(np.meshgrid(*[np.arange(side)] * 3, indexing="ij"), -1).reshape(-1, 3).astype(f32)
    return g[np.random.default_rng(seed).permutation(len(g))]


@pytest.mark.parametrize("min_dist_sq", [0.0, 0.3, 1.5])
def test_delete_on_a_lattice_ties_and_min_dist(min_dist_sq):
    """Every query has several points at exactly the same distance and MinDistSq cuts the search
    short: which id comes back depends on the patched tree's shape.  Must equal the oracle's."""
    pts = _lattice(12, 3)
    n = len(pts)
    t = kdtree.New(pts).With(MinDistSq=min_dist_sq) if min_dist_sq else kdtree.New(pts)
    o = O.KDTree(pts, min_dist_sq)
    rng = np.random.default_rng(8)
    q = np.concatenate([rng.integers(0, 23, (3000, 3)).astype(f32) * f32(0.5),      # lattice + half points
                        rng.uniform(-1, 12, (1000, 3)).astype(f32)])
    gone = rng.permutation(n)[: n // 2]
    for part in np.array_split(gone, 4):
        t.DeletePoints(part)
        for i in part:
            o.delete_point(int(i))
        assert t.Tree() == o.tree()
        ids, dsq = t.NearestBatch(q, 2.0)
        oi, od = o.nearest_batch(q, 2.0)
        assert np.array_equal(dsq, od)
        assert np.array_equal(ids, oi)
        for k in range(0, 40):
            r = t.Range(q[k], 1.5)
            ri, rd = o.range(q[k], 1.5)
            assert sorted((x.ID, float(x.DistSq)) for x in r) == sorted(zip(ri.tolist(), rd.tolist()))
        offs, rid, rdsq = t.RangeBatch(q[:500], 1.5)
        assert np.diff(offs).tolist() == [len(o.range(qq, 1.5)[0]) for qq in q[:500]]


def test_delete_presorted_device_queries_use_the_patched_tree():
    """nq large enough for the Morton presort of the device entry point."""
    import torch
    pts = _lattice(16, 4)
    t, o = kdtree.New(pts), O.KDTree(pts)
    gone = np.arange(0, len(pts), 3)
    t.DeletePoints(gone)
    for i in gone:
        o.delete_point(int(i))
    q = (np.random.default_rng(2).integers(0, 31, (1 << 16, 3)).astype(f32) * f32(0.5))
    dq = torch.from_numpy(q).cuda()
    ids = torch.empty(len(q), dtype=torch.int32, device="cuda")
    dsq = torch.empty(len(q), dtype=torch.float32, device="cuda")
    t.NearestBatchDev(dq.data_ptr(), len(q), 3.0, ids.data_ptr(), dsq.data_ptr(), presort=True)
    torch.cuda.synchronize()
    oi, od = o.nearest_batch(q, 3.0)
    assert np.array_equal(ids.cpu().numpy(), oi) and np.array_equal(dsq.cpu().numpy(), od)


def test_delete_all_points_on_a_line(golden):
    """kdtree_test.go:731-751, then root == nil: {-1, maxRange^2} (kdtree.go:84-86)."""
    g = golden("ref_kdtree.json")["delete_on_line"]
    pts = np.array(g["points"], f32)
    t = kdtree.New(pts)
    for i in range(len(pts)):
        t.DeletePoint(i)
        assert t.Nearest(pts[i], g["max_range"]).ID < 0
    assert t.LiveCount() == 0 and t.MaxDepth() == 0 and len(t.InOrder()) == 0
    n = t.Nearest(pts[0], 100.0)
    assert n.ID == -1 and n.DistSq == f32(100.0) * f32(100.0)
    assert t.Range(pts[0], 100.0) == []
    with pytest.raises(icp.ErrNotEnoughPairs):
        icp.PointToPointEvaluator(icp.NearestPointCorresponder(10.0), MinPairs=1).Evaluate(t, pts)


@pytest.mark.parametrize("n,seed", [(100, 0), (100, 1), (60000, 2)])
def test_delete_third_then_nearest_equals_oracle_and_naive(n, seed):
    """kdtree_test.go:864-885 (delete a third, then Nearest == brute force), also at a size that
    takes the device build, with deletions in two batches (two rebuilds)."""
    rng = np.random.default_rng(seed)
    w = f32(10.0)
    pts = synth.uniform_cloud(n, float(w), 40 + seed)
    t, o = kdtree.New(pts), O.KDTree(pts)
    gone = rng.permutation(n)[: n // 3]
    q = synth.uniform_cloud(2000, float(w), 50 + seed)
    for part in (gone[: len(gone) // 2], gone[len(gone) // 2:]):
        t.DeletePoints(part)
        for i in part:
            o.delete_point(int(i))
        ids, dsq = t.NearestBatch(q, 3.0)
        oi, od = o.nearest_batch(q, 3.0)
        assert np.array_equal(dsq, od)
        assert np.array_equal(ids, oi)
    assert t.LiveCount() == n - len(gone)
    assert not np.isin(ids, gone).any()
    keep = np.setdiff1d(np.arange(n), gone)
    for k in range(50):  # brute force over the remaining points
        i, d = O.naive_nearest(pts[keep], q[k], 3.0)
        assert ids[k] == (keep[i] if i >= 0 else -1) and dsq[k] == d
    assert sorted(t.InOrder().tolist()) == keep.tolist()


@pytest.mark.parametrize("min_dist_sq", [0.0, 0.6])
def test_delete_then_icp_pairs_on_a_lattice_equal_the_reference_walk(min_dist_sq):
    """Targets on lattice and half-lattice positions: most have several base points at exactly the
    same distance, so the pair the Corresponder reports depends on the patched tree's shape and the
    walk's visit order (correspondence.go:22-37 over kdtree.go:83-146).  Ids and DistSq must be the
    oracle's, pair for pair; so must the float64 sums of one Evaluate and a whole strict Fit."""
    pts = _lattice(14, 6)
    n = len(pts)
    t = kdtree.New(pts).With(MinDistSq=min_dist_sq) if min_dist_sq else kdtree.New(pts)
    o = O.KDTree(pts, min_dist_sq)
    gone = np.random.default_rng(12).permutation(n)[: n // 3]
    t.DeletePoints(gone)
    for i in gone:
        o.delete_point(int(i))
    rng = np.random.default_rng(13)
    target = rng.integers(0, 27, (20000, 3)).astype(f32) * f32(0.5)
    b, ti, d = icp.NearestPointCorresponder(MaxDist=1.6).PairsArrays(t, target)
    ob, ot, od = O.icp_pairs(o, target, 1.6)
    assert np.array_equal(ti, ot) and np.array_equal(b, ob) and np.array_equal(d, od)
    assert not np.isin(b, gone).any()
    e = icp.PointToPointEvaluator(icp.NearestPointCorresponder(MaxDist=1.6), MinPairs=6)
    ev = e.Evaluate(t, target)   # default: the reference's sums
    o32 = O.icp_evaluate(o, target, 1.6, 6, sums_mode=0)
    assert ev.NumPairs == o32["npairs"] and ev.Value == o32["value"] and np.array_equal(ev.Gradient, o32["gradient"])
    e.SumsMode = icp.SumsF64Tree
    ev = e.Evaluate(t, target)
    oe = O.icp_evaluate(o, target, 1.6, 6, sums_mode=1)
    assert ev.NumPairs == oe["npairs"]
    assert abs(float(ev.Value) - float(oe["value"])) <= 1e-6 * float(oe["value"])
    assert np.allclose(ev.Gradient, oe["gradient"], rtol=1e-5, atol=1e-6)


def test_delete_then_strict_fit_is_the_reference_fit_bit_for_bit():
    """A whole Fit (icp.go:24-72) on a handle with deletions, strict sums: the pose after every
    iteration depends on every pair, so equality of the final transform pins the patched walk."""
    c = synth.c4_icp(n=8000, width=1.9)
    t, o = kdtree.New(c["base"]), O.KDTree(c["base"])
    gone = np.arange(0, 8000, 4)
    t.DeletePoints(gone)
    for i in gone:
        o.delete_point(int(i))
    sess = icp.IcpSession(t, c["target"], c["max_dist"], c["min_pairs"], c["weight"], c["threshold"],
                          c["max_iteration"])   # default sums: the reference's
    for _ in range(c["max_iteration"]):
        sess.step()
    trans, stat, conv = sess.result()
    sess.close()
    of = O.icp_fit(o, c["target"], c["max_dist"], c["min_pairs"], c["weight"], c["threshold"],
                   c["max_iteration"], sums_mode=0)
    assert stat.NumIteration == of["num_iteration"]
    assert np.array_equal(trans, of["trans"])
    assert stat.Evaluated.Value == of["value"] and np.array_equal(stat.Evaluated.Gradient, of["gradient"])


def test_delete_then_icp_uses_remaining_points():
    c = synth.c4_icp(n=20000, width=2.7)
    t, o = kdtree.New(c["base"]), O.KDTree(c["base"])
    gone = np.arange(0, 20000, 3)
    t.DeletePoints(gone)
    for i in gone:
        o.delete_point(int(i))
    e = icp.PointToPointEvaluator(icp.NearestPointCorresponder(MaxDist=c["max_dist"]), MinPairs=6)
    ev = e.Evaluate(t, c["target"])   # default: the reference's sums
    o32 = O.icp_evaluate(o, c["target"], c["max_dist"], 6, sums_mode=0)
    assert ev.NumPairs == o32["npairs"] and ev.Value == o32["value"] and np.array_equal(ev.Gradient, o32["gradient"])
    e.SumsMode = icp.SumsF64Tree
    ev = e.Evaluate(t, c["target"])
    oe = O.icp_evaluate(o, c["target"], c["max_dist"], 6, sums_mode=1)
    assert ev.NumPairs == oe["npairs"]
    assert abs(float(ev.Value) - float(oe["value"])) <= 1e-6 * float(oe["value"])
    assert np.allclose(ev.Gradient, oe["gradient"], rtol=1e-5, atol=1e-8)


def test_many_delete_query_cycles_with_an_open_session():
    """Every cycle patches the tree further and uploads it again; a session created on a handle that
    already had deletions walks the handle's current patched tree, like an Evaluate call on the
    reference's KDTree would."""
    n = 40000
    pts = synth.uniform_cloud(n, 5.0, 91)
    q = synth.uniform_cloud(500, 5.0, 92)
    t, o = kdtree.New(pts), O.KDTree(pts)
    t.DeletePoints(np.arange(0, 100))
    for i in range(100):
        o.delete_point(i)
    early = icp.IcpSession(kdtree.New(pts), q, 1.0, 6)  # a handle without deletions: the implicit tree
    sess = icp.IcpSession(t, q, 1.0, 6, SumsMode=icp.SumsF64Tree)  # on the patched tree without points 0..99
    sess.partials()
    sums_before = sess.read_sums()
    assert np.allclose(sums_before, O.icp_evaluate(o, q, 1.0, 6, sums_mode=1)["raw10"], rtol=1e-12, atol=0)
    rng = np.random.default_rng(5)
    for cycle in range(25):
        gone = rng.choice(n, 200, replace=False)
        t.DeletePoints(gone)
        for i in gone:
            o.delete_point(int(i))
        ids, dsq = t.NearestBatch(q, 1.0)
        oi, od = o.nearest_batch(q, 1.0)
        assert np.array_equal(ids, oi) and np.array_equal(dsq, od), cycle
    sess.set_pose(np.eye(4, dtype=f32).reshape(-1), 0)
    sess.partials()
    now = sess.read_sums()
    want = O.icp_evaluate(o, q, 1.0, 6, sums_mode=1)["raw10"]
    assert now[9] == want[9] and not np.array_equal(now, sums_before)  # it saw the later deletions
    assert np.allclose(now, want, rtol=1e-12, atol=0)
    early.partials()
    assert early.read_sums()[9] == 500
    early.close()
    sess.close()


def _dump(t):
    import ctypes as C
    from pcgol_amd import _lib as L
    n = C.c_int64()
    L.check(L.lib().pcgx_kdtree_dump(t._h, None, 0, C.byref(n)))
    d = np.empty((max(n.value, 1), 4), np.int64)
    L.check(L.lib().pcgx_kdtree_dump(t._h, L.ptr(d), n.value, C.byref(n)))
    return d[: n.value]


@pytest.mark.parametrize("threads", ["1", "8"])
def test_batched_deletions_on_host_threads_leave_the_reference_tree(threads, monkeypatch):
    """A batch of deletions is carried out in call order -- on several host threads where the order allows it
    (csrc/knn_explicit.hip, xtree_delete_batch: deletions below different depth-6 nodes commute, a deletion above
    the cut runs alone in its place).  The batch holds the root, nodes above the cut, duplicates and ids deleted
    before; the tree afterwards is the oracle's after the same DeletePoint calls one by one, node for node (pre-order
    dump: id, dim, children), and so are Nearest's answers."""
    monkeypatch.setenv("PCGX_HOST_THREADS", threads)
    monkeypatch.setenv("PCGX_DELETE_PARALLEL_MIN", "64")
    n = 150_000
    pts = synth.uniform_cloud(n, 7.0, 23)
    t, o = kdtree.New(pts), O.KDTree(pts)
    rng = np.random.default_rng(29)
    d0 = _dump(t)
    assert np.array_equal(d0, o.dump())
    # the ids of the first 40 nodes in pre-order (the root, and nodes of the levels above the cut) go into the batches
    top = d0[:40, 0]
    first = np.concatenate([rng.permutation(n)[:20_000], top[:10], rng.integers(0, n, 500)])
    rng.shuffle(first)
    second = np.concatenate([rng.permutation(n)[:15_000], top[10:], first[:300]])
    rng.shuffle(second)
    for batch in (first, second):
        t.DeletePoints(batch)
        for i in batch:
            o.delete_point(int(i))
        assert np.array_equal(_dump(t), o.dump())
    gone = np.unique(np.concatenate([first, second]))
    assert t.LiveCount() == n - len(gone)
    q = synth.uniform_cloud(20_000, 7.0, 31)
    gi, gd = t.NearestBatch(q, 0.5)
    oi, od = o.nearest_batch(q, 0.5)
    assert np.array_equal(gi, oi) and np.array_equal(gd, od)
