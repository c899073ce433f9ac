"""GPU parity: ICP correspondence / evaluate / update / Fit (through the C ABI)
vs the CPU oracle and the reference's known-answer tables.
Tolerance: transform within 1e-5 absolute (BASELINE.json north_star)."""
import os

import numpy as np
import pytest

import oracle as O
from pcgol_amd import icp, kdtree, mat, synth

pytestmark = pytest.mark.gpu
f32 = np.float32
TOL = 1e-5
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_corresponder_golden(golden):
    g = golden("ref_icp.json")["corresponder"]
    t = kdtree.New(np.array(g["base"], f32))
    pairs = icp.NearestPointCorresponder(MaxDist=g["max_dist"]).Pairs(t, np.array(g["targets"], f32))
    assert [[p.BaseID, p.TargetID, float(p.SquaredDistance)] for p in pairs] == g["expected_pairs"]


def test_evaluator_golden(golden):
    g = golden("ref_icp.json")["evaluator"]
    base = np.array(g["base"], f32)
    delta = np.array(g["delta"], f32)
    target = base[g["target_base_ids"]] + delta
    t = kdtree.New(base)
    e = icp.PointToPointEvaluator(icp.NearestPointCorresponder(MaxDist=g["max_dist"]), MinPairs=g["min_pairs"])
    assert e.HasGradient() and not e.HasHessian()
    ev = e.Evaluate(t, target)
    assert ev.Value == f32(g["expected_value"])  # exact, evaluator_test.go:40-42
    fct = f32(g["step_factor"])
    dR = mat.RodriguesToRotation(ev.Gradient[3:] * fct)
    assert e.Evaluate(t, mat.Transform(dR, target)).Value < ev.Value
    assert e.Evaluate(t, target + ev.Gradient[:3] * fct).Value < ev.Value
    oe = O.icp_evaluate(O.KDTree(base), target, g["max_dist"], g["min_pairs"])
    assert ev.Value == oe["value"] and np.array_equal(ev.Gradient, oe["gradient"]) and ev.DistRMS == oe["dist_rms"]
    with pytest.raises(icp.ErrNotEnoughPairs):
        icp.PointToPointEvaluator(icp.NearestPointCorresponder(MaxDist=g["max_dist"]), MinPairs=4).Evaluate(t, target)


def _delta(ops):
    m = None
    for op in ops:
        f = O.translate(*op[1:]) if op[0] == "trans" else O.rotate(*op[1:])
        m = f if m is None else O.mat4_mul(m, f)
    return m


def test_fit_poses_golden(golden):
    """icp_test.go:13-98: 2 bases x 14 poses, residual <= 0.05; and the same transform as the oracle."""
    g = golden("ref_icp.json")["fit"]
    idx = g["indices"]
    for name, base in g["bases"].items():
        base = np.array(base, f32)
        for ops in g["deltas"]:
            target = O.mat4_transform(_delta(ops), base[idx])
            t = kdtree.New(base, MinDistSq=g["min_dist_sq"])
            reg = icp.PointToPointICPGradient(icp.PointToPointEvaluator(
                icp.NearestPointCorresponder(MaxDist=g["max_dist"]), MinPairs=g["min_pairs"]))
            trans, stat = reg.Fit(t, target)
            moved = mat.Transform(trans, target)
            res = np.mean(np.sum((moved - base[idx]).astype(np.float64) ** 2, axis=1))
            assert res <= g["max_residual"], (name, ops, res)
            o = O.icp_fit(O.KDTree(base, min_dist_sq=g["min_dist_sq"]), target, g["max_dist"], g["min_pairs"])
            assert stat.NumIteration == o["num_iteration"]
            assert np.array_equal(trans, o["trans"]), (name, ops)  # the default sums are the reference's own


def test_updater_matches_oracle():
    rng = np.random.default_rng(3)
    for case in range(50):
        g = (rng.random(6, dtype=f32) - f32(0.5)) * f32(10.0 if case % 2 else 0.05)
        tr = O.mat4_mul(O.translate(*rng.random(3, dtype=f32)), O.rotate(0, 0, 1, float(rng.random())))
        it = int(rng.integers(0, 20))
        ev = icp.Evaluated()
        ev.Gradient = g
        u = icp.GradientDescentUpdaterFactory().New()
        u.i = it
        t1, conv = u.Update(tr, ev)
        t2, conv2, it2 = O.icp_update(tr, g, it)
        assert conv == conv2 and u.i == it2
        assert np.array_equal(t1.view(np.uint32), t2.view(np.uint32))


@pytest.mark.parametrize("n,max_dist,min_dist_sq", [(20000, 0.5, 0.0), (20000, 0.5, 0.0004), (100000, 0.3, 0.0)])
def test_evaluate_vs_oracle(n, max_dist, min_dist_sq):
    c = synth.c4_icp(n=n, width=10.0 * (n / 1e6) ** (1 / 3))
    t = kdtree.New(c["base"], MinDistSq=min_dist_sq)
    o = O.KDTree(c["base"], min_dist_sq=min_dist_sq)
    # the default Evaluate forms the reference's sequential float32 sums: identical to the Go-semantics oracle
    ev = icp.PointToPointEvaluator(icp.NearestPointCorresponder(MaxDist=max_dist), MinPairs=6).Evaluate(t, c["target"])
    o32 = O.icp_evaluate(o, c["target"], max_dist, 6, sums_mode=0)
    assert ev.NumPairs == o32["npairs"] and ev.Value == o32["value"] and ev.DistRMS == o32["dist_rms"]
    assert np.array_equal(ev.Gradient, o32["gradient"])
    ev = icp.PointToPointEvaluator(icp.NearestPointCorresponder(MaxDist=max_dist), MinPairs=6,
                                   SumsMode=icp.SumsF64Tree).Evaluate(t, c["target"])
    oe = O.icp_evaluate(o, c["target"], max_dist, 6, sums_mode=1)  # float64 sums of the same float32 terms
    assert ev.NumPairs == oe["npairs"]
    # the device sums in float64 (different association): agree to float32 rounding of the results
    assert abs(float(ev.Value) - float(oe["value"])) <= 2e-7 * max(1.0, abs(float(oe["value"])))
    assert np.max(np.abs(ev.Gradient - oe["gradient"])) <= 1e-6
    assert abs(float(ev.DistRMS) - float(oe["dist_rms"])) <= 1e-5
    # pairs: identical to the oracle's
    b, tid, d = icp.NearestPointCorresponder(MaxDist=max_dist).PairsArrays(t, c["target"])
    ob, ot, od = O.icp_pairs(o, c["target"], max_dist)
    assert np.array_equal(b, ob) and np.array_equal(tid, ot) and np.array_equal(d, od)


@pytest.mark.parametrize("n", [20000, 200000])
def test_fit_vs_oracle(n):
    """Scaled-down C4: same density as the 1M config; the drop-in Fit returns the transform of the
    oracle (sequential float32 sums = Go semantics) bit for bit; the float64-tree mode stays within 1e-5."""
    c = synth.c4_icp(n=n, width=10.0 * (n / 1e6) ** (1 / 3))
    t = kdtree.New(c["base"])
    reg = icp.PointToPointICPGradient(
        icp.PointToPointEvaluator(icp.NearestPointCorresponder(MaxDist=c["max_dist"]), MinPairs=c["min_pairs"]),
        icp.GradientDescentUpdaterFactory(Weight=c["weight"], Threshold=c["threshold"], MaxIteration=c["max_iteration"]))
    trans, stat = reg.Fit(t, c["target"])
    o = O.icp_fit(O.KDTree(c["base"]), c["target"], c["max_dist"], c["min_pairs"], c["weight"], c["threshold"],
                  c["max_iteration"])
    assert stat.NumIteration == o["num_iteration"] == 20
    assert np.array_equal(trans, o["trans"]) and stat.Evaluated.Value == o["value"]
    assert np.array_equal(stat.Evaluated.Gradient, o["gradient"])
    reg.Evaluator.SumsMode = icp.SumsF64Tree
    trans, stat = reg.Fit(t, c["target"])
    assert np.max(np.abs(trans - o["trans"])) <= TOL
    assert np.max(np.abs(stat.Evaluated.Gradient - o["gradient"])) <= TOL


def test_fit_not_enough_pairs():
    base = synth.uniform_cloud(1000, 1.0, 1)
    target = base[:50] + f32(100.0)
    reg = icp.PointToPointICPGradient(icp.PointToPointEvaluator(icp.NearestPointCorresponder(MaxDist=0.5)))
    with pytest.raises(icp.ErrNotEnoughPairs) as ei:
        reg.Fit(kdtree.New(base), target)
    assert ei.value.stat.NumIteration == 1  # icp.go:50-53
    assert np.array_equal(ei.value.trans, mat.Translate(0, 0, 0))


def test_session_matches_fit():
    c = synth.c4_icp(n=50000, width=3.7)
    t = kdtree.New(c["base"])
    s = icp.IcpSession(t, c["target"], c["max_dist"], c["min_pairs"], c["weight"], c["threshold"], c["max_iteration"])
    for _ in range(c["max_iteration"]):
        s.partials()
        s.update()
    trans, stat, conv = s.result()
    reg = icp.PointToPointICPGradient(
        icp.PointToPointEvaluator(icp.NearestPointCorresponder(MaxDist=c["max_dist"]), MinPairs=c["min_pairs"]),
        icp.GradientDescentUpdaterFactory(Weight=c["weight"], Threshold=c["threshold"], MaxIteration=c["max_iteration"]))
    trans2, stat2 = reg.Fit(t, c["target"])
    assert conv and stat.NumIteration == stat2.NumIteration
    assert np.array_equal(trans, trans2)  # deterministic reduction: bitwise reproducible


def test_sharded_icp_rccl_single_rank():
    """The N > 1 code path (partials -> RCCL all-reduce on the device -> update) on one GPU:
    a 1-rank nccl group; must equal the fused single-GPU loop bit for bit."""
    import os
    import torch
    import torch.distributed as dist
    from pcgol_amd.distributed import ShardedIcp
    import socket
    assert torch.cuda.is_available()
    torch.cuda.set_device(0)  # the CUDA/HIP runtime must be up before ProcessGroupNCCL counts devices
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        c = synth.c4_icp(n=50000, width=3.7)
        t = kdtree.New(c["base"])
        args = (t, c["target"], c["max_dist"], c["min_pairs"], c["weight"], c["threshold"], c["max_iteration"])
        a = ShardedIcp(*args, force_exchange=True)
        assert a.exchange
        ta, sa, ca = a.fit()
        b = ShardedIcp(*args, SumsMode=icp.SumsF64Tree)  # (one rank alone would form the reference's sums)
        tb, sb, cb = b.fit()
        torch.cuda.synchronize()
        assert ca and cb and sa.NumIteration == sb.NumIteration == 20
        assert np.array_equal(ta, tb)
        a.close()
        b.close()
    finally:
        dist.destroy_process_group()


def _cold_sums(t, target, c, trans, it, mode=icp.SumsF64Tree):
    """Sums of one evaluation at a given pose from a FRESH session: its match[] is invalid,
    so the walk runs without the previous-match pruning hint."""
    s = icp.IcpSession(t, target, c["max_dist"], c["min_pairs"], c["weight"], c["threshold"], c["max_iteration"],
                       SumsMode=mode)
    s.set_pose(trans, it)
    s.partials()
    out = s.read_sums()
    s.close()
    return out


@pytest.mark.parametrize("mode", [icp.SumsF64Tree, icp.SumsReference], ids=["f64-tree", "reference"])
def test_hinted_walk_equals_cold_walk_every_iteration(mode):
    """Iterations >= 1 seed the walk's pruning bound with the previous iteration's match
    (icp.hip load_query): the 10 sums must equal, bit for bit, those of a walk without hints."""
    c = synth.c4_icp(n=30000, width=3.1)
    t = kdtree.New(c["base"])
    s = icp.IcpSession(t, c["target"], c["max_dist"], c["min_pairs"], c["weight"], c["threshold"], c["max_iteration"],
                       SumsMode=mode)
    for it in range(8):
        trans, _, _ = s.result()
        s.partials()
        hinted = s.read_sums()
        cold = _cold_sums(t, c["target"], c, trans, it, mode)
        assert hinted[9] > 0
        assert np.array_equal(hinted.view(np.uint64), cold.view(np.uint64)), it
        s.update()


def test_hinted_walk_exact_ties():
    """Lattice base, targets on cell centres (8 equidistant base points each), pose held at the
    identity with iter = 1 so that hints are taken from a previous launch and every target still
    ties exactly.  Sums and pair lists must equal the cold walk's and the oracle's."""
    g = np.arange(20, dtype=np.float32) * f32(0.25)
    base = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3).astype(np.float32)
    rng = np.random.default_rng(5)
    target = (base[rng.choice(len(base), 3000, replace=False)] + f32(0.125)).astype(np.float32)
    c = dict(max_dist=0.5, min_pairs=6, weight=None, threshold=None, max_iteration=20)
    t = kdtree.New(base)
    ident = mat.Translate(0, 0, 0)
    s = icp.IcpSession(t, target, c["max_dist"], c["min_pairs"], c["weight"], c["threshold"], c["max_iteration"],
                       SumsMode=icp.SumsF64Tree)
    s.partials()                     # iteration 0: fills match[]
    first = s.read_sums()
    s.set_pose(ident, 1)             # iter > 0: the next launch takes hints from match[]
    s.partials()
    hinted = s.read_sums()
    cold = _cold_sums(t, target, c, ident, 1)
    assert np.array_equal(hinted.view(np.uint64), cold.view(np.uint64))
    assert np.array_equal(hinted.view(np.uint64), first.view(np.uint64))  # identity re-projection is exact
    o = O.icp_evaluate(O.KDTree(base), target, 0.5, 6, sums_mode=1)
    assert int(hinted[9]) == o["npairs"] == len(target)
    ev = icp.FinishEvaluate(hinted, MinPairs=6)
    assert abs(float(ev.Value) - float(o["value"])) <= 2e-7 * max(1.0, abs(float(o["value"])))
    assert np.max(np.abs(ev.Gradient - o["gradient"])) <= 1e-6


def test_c4_full_size_vs_oracle():
    """BASELINE config C4 at full size (1M x 1M, 20 iterations): pairs of the first evaluation
    identical to the oracle's; the drop-in Fit (default sums = the reference's sequential float32
    additions) returns the Go-semantics oracle's transform and Evaluated bit for bit -- inside
    north_star's 1e-5 with room to spare.  The float64-tree mode (what a sharded sum computes) is
    within 1e-6 of the oracle run with float64 sums; its distance from the reference's float32 chain
    is that chain's own rounding noise, measured 1.6e-5 at this size -- which is why it is not the
    default."""
    c = synth.c4_icp()
    t = kdtree.New(c["base"])
    o = O.KDTree(c["base"])
    b, tid, d = icp.NearestPointCorresponder(MaxDist=c["max_dist"]).PairsArrays(t, c["target"])
    ob, ot, od = O.icp_pairs(o, c["target"], c["max_dist"])
    assert np.array_equal(b, ob) and np.array_equal(tid, ot) and np.array_equal(d, od)
    reg = icp.PointToPointICPGradient(
        icp.PointToPointEvaluator(icp.NearestPointCorresponder(MaxDist=c["max_dist"]), MinPairs=c["min_pairs"]),
        icp.GradientDescentUpdaterFactory(Weight=c["weight"], Threshold=c["threshold"], MaxIteration=c["max_iteration"]))
    trans, stat = reg.Fit(t, c["target"])
    o32 = O.icp_fit(o, c["target"], c["max_dist"], c["min_pairs"], c["weight"], c["threshold"], c["max_iteration"],
                    sums_mode=0)
    assert stat.NumIteration == o32["num_iteration"] == 20
    assert np.max(np.abs(trans - o32["trans"])) <= 1e-5   # north_star's bar, on the default path
    assert np.array_equal(trans, o32["trans"]) and stat.Evaluated.Value == o32["value"]   # in fact identical
    assert np.array_equal(stat.Evaluated.Gradient, o32["gradient"]) and stat.Evaluated.DistRMS == o32["dist_rms"]
    reg.Evaluator.SumsMode = icp.SumsF64Tree
    trans64, stat64 = reg.Fit(t, c["target"])
    o64 = O.icp_fit(o, c["target"], c["max_dist"], c["min_pairs"], c["weight"], c["threshold"], c["max_iteration"],
                    sums_mode=1)
    assert stat64.NumIteration == o64["num_iteration"] == 20
    assert np.max(np.abs(trans64 - o64["trans"])) <= 1e-6
    assert np.max(np.abs(trans64 - o32["trans"])) <= 3e-5   # the reference chain's own rounding noise (not the default path)


@pytest.mark.parametrize("n,max_dist,grid", [(5000, 0.5, "1"), (20003, 0.03, "1"), (200000, 0.5, "1"), (20003, 0.03, "0")],
                         ids=["5000", "20003-no-partner", "200000", "20003-walk-only"])
def test_strict_mode_is_bit_identical_to_the_reference_sums(n, max_dist, grid, monkeypatch):
    """STRICT sums (sequential float32 in target order, evaluator.go:122-145, evaluated in parallel:
    csrc/strict_sum.h): Evaluated and every pose of the Fit loop equal the oracle's Go-semantics
    run bit for bit -- also when many targets find no partner (max_dist 0.03) and the count is no
    multiple of the tile size; and when every pair comes from the tree walk (PCGX_GRID=0: the pairs
    reach the caller-order copy the sums read through the walk kernel's stores)."""
    monkeypatch.setenv("PCGX_GRID", grid)
    c = synth.c4_icp(n=n, width=10.0 * (n / 1e6) ** (1 / 3))
    c["max_dist"] = max_dist
    if max_dist < 0.1:  # targets that never find a partner, scattered through the target order
        far = c["target"][::37] + f32(50.0)
        c["target"] = np.ascontiguousarray(np.insert(c["target"], np.arange(0, len(far)) * 30, far, axis=0))
    t, o = kdtree.New(c["base"]), O.KDTree(c["base"])
    s = icp.IcpSession(t, c["target"], c["max_dist"], c["min_pairs"], c["weight"], c["threshold"], c["max_iteration"])
    s.set_strict(True)
    trans = O.translate(0, 0, 0)
    it = 0
    tt = c["target"].copy()
    for k in range(c["max_iteration"]):
        s.step()
        tr, st, conv = s.result()
        oe = O.icp_evaluate(o, tt, c["max_dist"], c["min_pairs"], sums_mode=0)
        assert st.Evaluated.Value == oe["value"] and st.Evaluated.DistRMS == oe["dist_rms"], k
        assert np.array_equal(st.Evaluated.Gradient, oe["gradient"]), k
        trans, oconv, it = O.icp_update(trans, oe["gradient"], it, c["weight"], c["threshold"], c["max_iteration"])
        assert np.array_equal(tr, trans) and conv == oconv, k
        tt = O.mat4_transform(trans, c["target"]) if n <= 5000 else synth.transform_points(trans, c["target"])
    assert 6 <= st.Evaluated.NumPairs <= len(c["target"]) and (max_dist > 0.1 or st.Evaluated.NumPairs < len(c["target"]))
    s.close()


def test_strict_mode_c4_full_size():
    """C4 at full size: with strict sums the Fit equals the reference-semantics oracle bit for bit
    (the default float64 sums differ from it by the reference's own rounding noise, 1.6e-5)."""
    c = synth.c4_icp()
    t = kdtree.New(c["base"])
    s = icp.IcpSession(t, c["target"], c["max_dist"], c["min_pairs"], c["weight"], c["threshold"], c["max_iteration"])
    s.set_strict(True)
    for _ in range(c["max_iteration"]):
        s.step()
    tr, st, conv = s.result()
    s.close()
    o32 = O.icp_fit(O.KDTree(c["base"]), c["target"], c["max_dist"], c["min_pairs"], c["weight"], c["threshold"],
                    c["max_iteration"], sums_mode=0)
    assert conv and st.NumIteration == o32["num_iteration"] == 20
    assert np.array_equal(tr, o32["trans"])
    assert st.Evaluated.Value == o32["value"] and np.array_equal(st.Evaluated.Gradient, o32["gradient"])


@pytest.mark.parametrize("n,max_dist", [(3, 0.5), (2048, 0.5), (70001, 0.5), (300000, 0.04)])
def test_strict_parallel_equals_the_one_wave_chain(n, max_dist, monkeypatch):
    """The parallel evaluation of the sequential float32 sums (set_strict 1: class summaries,
    composition, exact recomputation where a record does not cover the state) against the plain
    dependent chain on the device (set_strict 2), sum by sum, over a whole Fit.  With
    PCGX_STRICT_SELFCHECK every step of the chain walk is re-derived term by term inside the
    kernel: none may differ."""
    monkeypatch.setenv("PCGX_STRICT_SELFCHECK", "1")
    c = synth.c4_icp(n=max(n, 64), width=10.0 * (max(n, 64) / 1e6) ** (1 / 3))
    target = c["target"][:n] if n < 64 else c["target"]
    if max_dist < 0.1:
        far = target[::11] + f32(50.0)
        target = np.ascontiguousarray(np.insert(target, np.arange(0, len(far)) * 9, far, axis=0))
    t = kdtree.New(c["base"])
    a = icp.IcpSession(t, target, max_dist, 1, c["weight"], c["threshold"], c["max_iteration"])
    b = icp.IcpSession(t, target, max_dist, 1, c["weight"], c["threshold"], c["max_iteration"])
    a.set_strict(1)
    b.set_strict(2)
    for k in range(c["max_iteration"]):
        a.step()
        b.step()
        sa, sb = a.read_sums(), b.read_sums()
        assert np.array_equal(sa.view(np.uint64), sb.view(np.uint64)), (k, sa, sb)
        st = a.strict_stats()
        assert not st[12:16].any() and st[6] == 0 and st[7] == 0, (k, st[:16])
    ta, sta, _ = a.result()
    tb, stb, _ = b.result()
    assert np.array_equal(ta, tb) and sta.Evaluated.NumPairs == stb.Evaluated.NumPairs
    a.close()
    b.close()


def test_strict_sums_on_structured_terms():
    """Targets on a lattice that coincide with base points: every term is exact (zeros, ones), sums
    land exactly on binade boundaries -- the point records' case -- and a second cloud shifted by
    a power of two makes every addition of the translation gradient a tie."""
    g = np.arange(0, 40, dtype=np.float32) * f32(0.25)
    base = np.ascontiguousarray(np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3))
    t = kdtree.New(base)
    o = O.KDTree(base)
    for shift in (0.0, 0.0625):
        target = np.ascontiguousarray(base[::-1] + f32(shift))
        s = icp.IcpSession(t, target, 0.5, 6, np.full(6, 0.3, np.float32), np.full(6, -1.0, np.float32), 5)
        s.set_strict(1)
        s.step()
        _, st, _ = s.result()
        oe = O.icp_evaluate(o, target, 0.5, 6, sums_mode=0)
        assert st.Evaluated.Value == oe["value"] and st.Evaluated.DistRMS == oe["dist_rms"]
        assert np.array_equal(st.Evaluated.Gradient, oe["gradient"])
        s.close()


@pytest.mark.parametrize("tables", [True, False])
def test_strict_sums_that_hover_around_zero(monkeypatch, tables):
    """Targets = base points + zero-mean noise, no transform: all six gradient sums wander around zero
    from the first pair to the last -- sign changes and three binades inside one tile, so many tiles
    have no window at all and go through their leaves' records (runs of leaves under equal windows,
    the rest added up term by term: the helper waves of the chain kernel).  Still the oracle's
    Go-semantics sums, bit for bit; with PCGX_STRICT_SELFCHECK every step of the walk is re-derived term
    by term inside the kernel as well (the device-only paths the host model does not mirror).
    tables: the job kernel's candidate tables (a tile carried out from the start states around its guess: the
    walker looks its state up) are in use, or switched off (PCGX_STRICT_NOSPEC) so that every such tile goes
    through its leaves' records."""
    monkeypatch.setenv("PCGX_STRICT_SELFCHECK", "1")
    if not tables:
        monkeypatch.setenv("PCGX_STRICT_NOSPEC", "1")
    n = 200_000
    rng = np.random.Generator(np.random.PCG64(77))
    base = synth.uniform_cloud(n, 10.0 * (n / 1e6) ** (1 / 3), 2)
    noise = ((rng.integers(0, 1 << 16, size=(n, 3)).astype(np.float32) / f32(1 << 16)) - f32(0.5)) * f32(0.02)
    target = np.ascontiguousarray((base[rng.permutation(n)] + noise).astype(np.float32))
    t, o = kdtree.New(base), O.KDTree(base)
    s = icp.IcpSession(t, target, 0.5, 6, np.full(6, 0.3, np.float32), np.full(6, -1.0, np.float32), 3)
    s.set_strict(1)
    trans = O.translate(0, 0, 0)
    it = 0
    tt = target.copy()
    resolved = looked_up = 0
    for k in range(3):
        s.step()
        tr, st, conv = s.result()
        sst = s.strict_stats()
        resolved += int(sst[2])
        looked_up += int(sst[25]) + int(sst[45])
        assert not sst[12:16].any() and sst[6] == 0 and sst[7] == 0, (k, sst[:24])
        assert sst[16] > 0 or k > 0   # tiles without a window were recomputed from their leaf records
        oe = O.icp_evaluate(o, tt, 0.5, 6, sums_mode=0)
        assert st.Evaluated.Value == oe["value"] and st.Evaluated.DistRMS == oe["dist_rms"], k
        assert np.array_equal(st.Evaluated.Gradient, oe["gradient"]), k
        trans, oconv, it = O.icp_update(trans, oe["gradient"], it, np.full(6, 0.3, np.float32), np.full(6, -1.0, np.float32), 3)
        assert np.array_equal(tr, trans), k
        tt = synth.transform_points(trans, target)
    s.close()
    # the case does what it is for: many tiles whose record does not cover the state
    assert resolved + looked_up >= 20 and resolved >= 5, (resolved, looked_up)
    assert (looked_up > 0) == tables


@pytest.mark.parametrize("slots", ["0", "1"])
def test_strict_sums_when_the_job_slots_run_out(slots, monkeypatch):
    """Tiles that cross a level or have no window normally get a slot (leaf records + terms for the chain kernel's
    helpers).  With none / one per shard most of them find no slot: their records say so and the chain kernel forms
    their terms again from the pairs and adds all 2048 one after the other.  Same bits as the oracle, every step
    re-derived inside the kernel (PCGX_STRICT_SELFCHECK)."""
    monkeypatch.setenv("PCGX_STRICT_SLOTS_PER_SHARD", slots)
    monkeypatch.setenv("PCGX_STRICT_SELFCHECK", "1")
    n = 150_000
    rng = np.random.Generator(np.random.PCG64(78))
    base = synth.uniform_cloud(n, 10.0 * (n / 1e6) ** (1 / 3), 2)
    noise = ((rng.integers(0, 1 << 16, size=(n, 3)).astype(np.float32) / f32(1 << 16)) - f32(0.5)) * f32(0.02)
    target = np.ascontiguousarray((base[rng.permutation(n)] + noise).astype(np.float32))
    t, o = kdtree.New(base), O.KDTree(base)
    s = icp.IcpSession(t, target, 0.5, 6, np.full(6, 0.3, np.float32), np.full(6, -1.0, np.float32), 2)
    no_slot = 0
    trans = O.translate(0, 0, 0)
    it = 0
    tt = target.copy()
    for k in range(2):
        s.step()
        tr, st, conv = s.result()
        sst = s.strict_stats()
        no_slot += int(sst[5])
        assert not sst[12:16].any() and sst[6] == 0 and sst[7] == 0, (k, sst[:24])
        oe = O.icp_evaluate(o, tt, 0.5, 6, sums_mode=0)
        assert st.Evaluated.Value == oe["value"] and np.array_equal(st.Evaluated.Gradient, oe["gradient"]), k
        trans, oconv, it = O.icp_update(trans, oe["gradient"], it, np.full(6, 0.3, np.float32), np.full(6, -1.0, np.float32), 2)
        assert np.array_equal(tr, trans), k
        tt = synth.transform_points(trans, target)
    s.close()
    assert no_slot > 0   # the path was taken


@pytest.mark.parametrize("n", [1, 7, 2047, 2048, 2049, 4100])
def test_default_sums_on_small_and_ragged_targets(n):
    """Targets of fewer pairs than a tile, exactly a tile, one more: the padding behind the last pair carries -0.0f
    (x + (-0.0f) == x for every x), the default Evaluate / Fit return the oracle's bits."""
    base = synth.uniform_cloud(5000, 2.0, 31)
    rng = np.random.default_rng(n)
    target = (base[rng.choice(len(base), n, replace=n > len(base))] + f32(0.003)).astype(np.float32)
    t, o = kdtree.New(base), O.KDTree(base)
    ev = icp.PointToPointEvaluator(icp.NearestPointCorresponder(MaxDist=0.5), MinPairs=1).Evaluate(t, target)
    oe = O.icp_evaluate(o, target, 0.5, 1, sums_mode=0)
    assert ev.NumPairs == oe["npairs"] and ev.Value == oe["value"] and ev.DistRMS == oe["dist_rms"]
    assert np.array_equal(ev.Gradient, oe["gradient"])
    reg = icp.PointToPointICPGradient(icp.PointToPointEvaluator(icp.NearestPointCorresponder(MaxDist=0.5), MinPairs=1),
                                      icp.GradientDescentUpdaterFactory(MaxIteration=5))
    tr, st = reg.Fit(t, target)
    of = O.icp_fit(o, target, 0.5, 1, None, None, 5)
    assert st.NumIteration == of["num_iteration"] and np.array_equal(tr, of["trans"])


@pytest.mark.parametrize("wf", [icp.WeightConstant(0.25), icp.WeightInverse(0.01), icp.WeightHuber(0.0009),
                                icp.WeightTukey(0.004)], ids=["constant", "inverse", "huber", "tukey"])
def test_builtin_weight_fns_match_the_oracle(wf):
    """PointToPointEvaluator.WeightFn (evaluator.go:19-23,130): the built-in forms weigh every term
    in float32 exactly as the Go closure would.  Strict sums: Evaluated and the whole Fit equal the
    oracle bit for bit; the float64 reduction agrees with the oracle's float64 sums to rounding."""
    n = 40000
    c = synth.c4_icp(n=n, width=10.0 * (n / 1e6) ** (1 / 3))
    t, o = kdtree.New(c["base"]), O.KDTree(c["base"])
    O.set_weight_fn(wf.kind, wf.a)
    try:
        ev = icp.PointToPointEvaluator(icp.NearestPointCorresponder(MaxDist=c["max_dist"]), MinPairs=6, WeightFn=wf)
        got = ev.Evaluate(t, c["target"])   # default sums: the reference's
        e32 = O.icp_evaluate(o, c["target"], c["max_dist"], 6, sums_mode=0)
        assert got.Value == e32["value"] and np.array_equal(got.Gradient, e32["gradient"]) and got.DistRMS == e32["dist_rms"]
        ev.SumsMode = icp.SumsF64Tree
        got = ev.Evaluate(t, c["target"])
        exp = O.icp_evaluate(o, c["target"], c["max_dist"], 6, sums_mode=1)
        assert abs(got.Value - exp["value"]) <= 1e-6 * abs(exp["value"]) + 1e-12
        assert np.allclose(got.Gradient, exp["gradient"], rtol=2e-6, atol=1e-9)
        s = icp.IcpSession(t, c["target"], c["max_dist"], 6, c["weight"], c["threshold"], c["max_iteration"], WeightFn=wf)
        for _ in range(c["max_iteration"]):
            s.step()
        tr, st, conv = s.result()
        s.close()
        o32 = O.icp_fit(o, c["target"], c["max_dist"], 6, c["weight"], c["threshold"], c["max_iteration"], sums_mode=0)
        assert np.array_equal(tr, o32["trans"]) and st.Evaluated.Value == o32["value"]
        assert np.array_equal(st.Evaluated.Gradient, o32["gradient"]) and st.Evaluated.DistRMS == o32["dist_rms"]
        # the weights differ from 1 on this data (the test would pass trivially otherwise)
        w = np.array([wf(d) for d in np.linspace(1e-5, 0.01, 50, dtype=np.float32)])
        assert np.any(w != 1.0)
    finally:
        O.set_weight_fn(0, 0.0)


def test_custom_weight_closure_is_refused():
    with pytest.raises(NotImplementedError):
        icp.PointToPointEvaluator(icp.NearestPointCorresponder(MaxDist=1.0), WeightFn=lambda d: 1.0)


def test_more_than_two_to_the_24_pairs_default_sums_equal_the_oracle():
    """Above 2^24 pairs the reference's ninth sum (0 + 1 + 1 + ... in float32, evaluator.go:143) stops at 2^24: with
    the default weight the device does not chain that sum but takes min(pairs, 2^24) (csrc/strict.hip, chain kernel's
    ticket) -- pinned here against the oracle's sequential float32 sums with 17.3M pairs on one GPU: Evaluated bit
    for bit (its 1/sum-of-weights factor is 2^-24, not 1/pairs), and the pair count itself."""
    n_base, n_t = 100_000, 17_300_000
    base = synth.uniform_cloud(n_base, 4.0, 41)
    rng = np.random.default_rng(43)
    target = (base[rng.integers(0, n_base, n_t)] + rng.uniform(-0.004, 0.004, (n_t, 3)).astype(np.float32)).astype(np.float32)
    t, o = kdtree.New(base), O.KDTree(base)
    ev = icp.PointToPointEvaluator(icp.NearestPointCorresponder(MaxDist=1.0), MinPairs=6).Evaluate(t, target)
    oe = O.icp_evaluate(o, target, 1.0, 6, sums_mode=0)
    assert ev.NumPairs == oe["npairs"] == n_t > (1 << 24)
    assert ev.Value == oe["value"] and ev.DistRMS == oe["dist_rms"] and np.array_equal(ev.Gradient, oe["gradient"])
    # the stalled sum shows: Value is (sum of d^2) * 2^-24, above the mean squared distance by pairs / 2^24
    o64 = O.icp_evaluate(o, target, 1.0, 6, sums_mode=1)
    assert 1.02 < float(ev.Value) / float(o64["value"]) < 1.04


@pytest.mark.parametrize("wf", [icp.WeightInverse(0.01), icp.WeightHuber(0.0009)], ids=["inverse", "huber"])
def test_weight_fns_at_c4_full_size_match_the_oracle(wf):
    """The built-in weight forms at C4's full size (1M x 1M): nine chained sums (the sum of the weights is a chain
    of its own then), every iteration's Evaluated and the final pose equal to the Go-semantics oracle bit for bit."""
    c = synth.c4_icp()
    t, o = kdtree.New(c["base"]), O.KDTree(c["base"])
    O.set_weight_fn(wf.kind, wf.a)
    try:
        s = icp.IcpSession(t, c["target"], c["max_dist"], 6, c["weight"], c["threshold"], c["max_iteration"], WeightFn=wf)
        for _ in range(c["max_iteration"]):
            s.step()
        tr, st, conv = s.result()
        sst = s.strict_stats()
        s.close()
        o32 = O.icp_fit(o, c["target"], c["max_dist"], 6, c["weight"], c["threshold"], c["max_iteration"], sums_mode=0)
        assert st.NumIteration == o32["num_iteration"] == 20
        assert np.array_equal(tr, o32["trans"]) and st.Evaluated.Value == o32["value"]
        assert np.array_equal(st.Evaluated.Gradient, o32["gradient"]) and st.Evaluated.DistRMS == o32["dist_rms"]
        assert sst[63] == 0   # no workgroup of the summary kernel gave up waiting for the tiles before it
        assert sst[62] == 0   # no walker of the chain kernel gave up waiting for the chunk before its own
    finally:
        O.set_weight_fn(0, 0.0)


def test_chain_kernel_chunks_hand_the_walk_on():
    """2.6M targets = 1270 tiles per sum = three chunks of the chain kernel, each a workgroup of its own whose walker
    starts where the chunk before it ended (csrc/strict.hip, StrictWork::chunk_state): iteration 0's Evaluated against
    the oracle's sequential float32 sums bit for bit, then five iterations against the one-wave chain on the device;
    no walker gave up its wait (it would have walked the earlier chunks alone: same sums, no parallelism)."""
    n_base, n_t = 300_000, 2_600_000
    base = synth.uniform_cloud(n_base, 6.7, 51)
    rng = np.random.default_rng(53)
    target = synth.transform_points(synth.icp_pose(), base[rng.integers(0, n_base, n_t)] +
                                    rng.uniform(-0.01, 0.01, (n_t, 3)).astype(np.float32)).astype(np.float32)
    t, o = kdtree.New(base), O.KDTree(base)
    ev = icp.PointToPointEvaluator(icp.NearestPointCorresponder(MaxDist=0.5), MinPairs=6).Evaluate(t, target)
    oe = O.icp_evaluate(o, target, 0.5, 6, sums_mode=0)
    assert ev.NumPairs == oe["npairs"] > 2_500_000
    assert ev.Value == oe["value"] and ev.DistRMS == oe["dist_rms"] and np.array_equal(ev.Gradient, oe["gradient"])
    cfg = dict(MaxDist=0.5, MinPairs=6, Weight=np.full(6, 0.3, np.float32), Threshold=np.full(6, -1.0, np.float32), MaxIteration=20)
    a, b = icp.IcpSession(t, target, **cfg), icp.IcpSession(t, target, **cfg)
    a.set_strict(1)
    b.set_strict(2)
    for k in range(5):
        a.step()
        b.step()
        assert np.array_equal(a.read_sums().view(np.uint64), b.read_sums().view(np.uint64)), k
        st = a.strict_stats()
        assert st[62] == 0 and st[63] == 0, (k, st[60:64])
    a.close()
    b.close()


def test_pairs_kept_on_certificates_are_the_searched_ones():
    """From a Fit's second iteration on a target keeps last iteration's partner when its DistSq to it is below the
    partner's certificate (csrc/icp.hip, icp_grid_kernel; csrc/knn_grid.hip, grid_cert_kernel) -- no search.  On a cloud
    with twins and a lattice patch (exact ties: never certified): the share that is kept grows over the iterations,
    and the Fit is the oracle's bit for bit (every kept pair is the pair a search would have returned)."""
    rng = np.random.default_rng(3)
    base = np.concatenate([
        synth.uniform_cloud(60_000, 4.0, 21),
        np.stack(np.meshgrid(*[np.arange(10, dtype=np.float32) * np.float32(0.0625) + np.float32(1.0)] * 3), -1).reshape(-1, 3),
    ]).astype(np.float32)
    base = np.ascontiguousarray(np.concatenate([base, base[:200]]))                  # twins
    target = synth.transform_points(synth.icp_pose(), base[rng.permutation(len(base))[:50_000]])
    w, th = np.full(6, 0.3, np.float32), np.full(6, -1.0, np.float32)
    t = kdtree.New(base)
    s = icp.IcpSession(t, target, 0.5, 6, w, th, 12)
    kept = []
    for it in range(12):
        kept.append(s.grid_stats()[5])
        s.step()
    trans, st, _ = s.result()
    assert kept[0] == 0 and kept[1] > 0.3 * len(target) and kept[-1] > 0.9 * len(target), kept
    o = O.icp_fit(O.KDTree(base), target, 0.5, 6, w, th, 12, sums_mode=0)
    assert st.NumIteration == o["num_iteration"] == 12
    assert np.array_equal(np.asarray(trans).ravel(), np.asarray(o["trans"]).ravel())
    assert np.float32(st.Evaluated.Value) == o["value"]
    assert np.array_equal(np.asarray(st.Evaluated.Gradient, np.float32), o["gradient"])


def test_a_step_without_the_leftover_walk_is_enqueued_again_when_the_grid_leaves_a_target():
    """From a Fit's second Evaluate on the leftover walk is not launched behind a strict session's grid pass (csrc/icp.hip,
    enqueue_corr: it finds nothing to do, and its launch is 5 us of a 70 us step).  A target the grid cannot answer then
    stops the step on the device, and settle() enqueues it again with the walk.  Forced here (PCGX_TEST_ICP_FORCE_WALK:
    every 1000th target goes to the walk from the second Evaluate on), in a process of its own (the knob is read once):
    the Fit is the oracle's bit for bit, and the trace line says that steps were enqueued again."""
    import subprocess, sys, textwrap
    code = textwrap.dedent("""
        import numpy as np
        import oracle as O
        from pcgol_amd import icp, kdtree, synth
        n = 120_000
        c = synth.c4_icp(n=n, width=10.0 * (n / 1e6) ** (1 / 3))
        reg = icp.PointToPointICPGradient(
            icp.PointToPointEvaluator(icp.NearestPointCorresponder(MaxDist=c["max_dist"]), MinPairs=c["min_pairs"]),
            icp.GradientDescentUpdaterFactory(Weight=c["weight"], Threshold=c["threshold"], MaxIteration=c["max_iteration"]))
        t = kdtree.New(c["base"])
        trans, st = reg.Fit(t, c["target"])
        o = O.icp_fit(O.KDTree(c["base"]), c["target"], c["max_dist"], c["min_pairs"], c["weight"], c["threshold"], c["max_iteration"], sums_mode=0)
        assert st.NumIteration == o["num_iteration"] == 20
        assert np.array_equal(np.asarray(trans).ravel(), np.asarray(o["trans"]).ravel())
        assert np.float32(st.Evaluated.Value) == o["value"]
        assert np.array_equal(np.asarray(st.Evaluated.Gradient, np.float32), o["gradient"])
        # a session stepped by hand, looked at in the middle (read_sums) and at the end
        s = icp.IcpSession(t, c["target"], c["max_dist"], c["min_pairs"], c["weight"], c["threshold"], c["max_iteration"])
        for _ in range(7):
            s.step()
        s.read_sums()
        for _ in range(13):
            s.step()
        tr2, st2, _ = s.result()
        assert st2.NumIteration == 20 and np.array_equal(np.asarray(tr2).ravel(), np.asarray(o["trans"]).ravel())
        # a Fit picked up in its middle (set_pose with the updater's count > 0: the device's NumIteration restarts at 0,
        # and settle() must count the Evaluates enqueued since THAT write, not the updater's iterations -- ADVICE r5):
        # five iterations, the pose carried into a fresh session, fifteen more = the oracle's twenty
        s3 = icp.IcpSession(t, c["target"], c["max_dist"], c["min_pairs"], c["weight"], c["threshold"], c["max_iteration"])
        for _ in range(5):
            s3.step()
        tr5, st5, _ = s3.result()
        assert st5.NumIteration == 5
        s4 = icp.IcpSession(t, c["target"], c["max_dist"], c["min_pairs"], c["weight"], c["threshold"], c["max_iteration"])
        s4.set_pose(np.asarray(tr5, np.float32), 5)
        for _ in range(15):
            s4.step()
        tr4, st4, _ = s4.result()
        assert st4.NumIteration == 15, st4.NumIteration
        assert np.array_equal(np.asarray(tr4).ravel(), np.asarray(o["trans"]).ravel())
        assert np.float32(st4.Evaluated.Value) == o["value"]
        # ... and the same session turned back in its middle: set_pose behind steps that were speculated on
        s4.set_pose(np.asarray(tr5, np.float32), 5)
        for _ in range(15):
            s4.step()
        s4.set_strict(True)   # (settles what is pending before the mode could change)
        tr4b, st4b, _ = s4.result()
        assert st4b.NumIteration == 15 and np.array_equal(np.asarray(tr4b).ravel(), np.asarray(o["trans"]).ravel())
        print("fit ok")
    """)
    env = dict(os.environ, PCGX_TEST_ICP_FORCE_WALK="1000", PCGX_ICP_SPEC_TRACE="1")
    r = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r)\n" % ROOT + code], env=env, cwd=ROOT, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0 and "fit ok" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
    assert r.stderr.count("enqueued again") == 4, r.stderr[-3000:]   # once per session: the walk stays on afterwards
