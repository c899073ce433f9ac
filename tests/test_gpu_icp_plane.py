"""GPU checks of the point-to-plane / Gauss-Newton ICP extension (6x6 normal equations).

NO REFERENCE PARITY EXISTS for this path: seqsense/pcgol has no point-to-plane evaluator and
never writes Evaluated.Hessian (evaluator.go:28,76).  The HIP path is checked against the CPU
oracle's independent float64 restatement of the extension's definition
(oracle/plane_oracle.c, "parity unpinned") and against synthetic ground truth.
Tolerances: sums 1e-11 relative to the sum of absolute terms (float64 accumulation order
differs); transform within 1e-5 absolute (the tolerance BASELINE.json states for ICP)."""
import numpy as np
import pytest

import oracle as O
from pcgol_amd import icp, kdtree, mat, synth

pytestmark = pytest.mark.gpu
f32 = np.float32
TOL = 1e-5


def _scene(n, seed=0):
    c = synth.c4_plane(n, base_seed=6 + seed, perm_seed=7 + seed)
    return c


def test_plane_sums_match_oracle():
    c = _scene(50_000)
    t = kdtree.New(c["base"])
    ev = icp.PointToPlaneEvaluator(icp.NearestPointCorresponder(MaxDist=c["max_dist"]), c["normals"], MinPairs=6)
    assert ev.HasGradient() and ev.HasHessian()
    sums = ev.Sums(t, c["target"])
    osums = O.plane_sums(O.KDTree(c["base"]), c["normals"], c["target"], c["max_dist"])
    assert sums[29] == osums[29] and sums[28] == osums[28]  # pair count / weight: exact
    scale = np.maximum(np.abs(osums), 1e-30)
    # every term is the same float32 value on both sides; only the float64 summation order differs
    assert np.all(np.abs(sums - osums) <= 1e-11 * np.maximum(scale, osums[29]))
    e = ev.Evaluate(t, c["target"])
    oe = O.plane_finish(osums, 6)
    assert abs(float(e.Value) - float(oe["value"])) <= 1e-6 * float(oe["value"])
    assert np.allclose(e.Gradient, oe["gradient"], rtol=1e-5, atol=1e-9)
    assert np.allclose(e.Hessian, oe["hessian"], rtol=1e-5, atol=1e-9)
    assert np.array_equal(e.Hessian.reshape(6, 6), e.Hessian.reshape(6, 6).T)


def test_plane_fit_matches_oracle_and_ground_truth():
    c = _scene(40_000)
    t = kdtree.New(c["base"])
    reg = icp.PointToPlaneICP(
        icp.PointToPlaneEvaluator(icp.NearestPointCorresponder(MaxDist=c["max_dist"]), c["normals"], MinPairs=6),
        icp.GaussNewtonUpdaterFactory(Threshold=c["threshold"], MaxIteration=8))
    trans, stat = reg.Fit(t, c["target"])
    o = O.plane_fit(O.KDTree(c["base"]), c["normals"], c["target"], c["max_dist"], 6, c["threshold"], 0.0, 8)
    assert stat.NumIteration == o["num_iteration"] == 8
    assert np.max(np.abs(trans - o["trans"])) <= TOL
    # ground truth: target = T * base[perm], so Fit must recover T^-1 (Gauss-Newton converges
    # quadratically; the point-to-point gradient updater of the reference needs far more iterations)
    inv = np.linalg.inv(synth.icp_pose().astype(np.float64).reshape(4, 4).T).T.reshape(-1)
    assert np.max(np.abs(trans.astype(np.float64) - inv)) <= 2e-4
    moved = mat.Transform(trans, c["target"])
    perm = np.random.Generator(np.random.PCG64(7)).permutation(len(c["base"]))
    assert np.sqrt(np.mean(np.sum((moved - c["base"][perm]) ** 2, axis=1))) < 2e-3
    assert float(stat.Evaluated.Value) < 1e-7
    assert stat.Evaluated.Hessian.shape == (36,) and np.all(np.diag(stat.Evaluated.Hessian.reshape(6, 6)) > 0)


def test_plane_session_steps_equal_fit_and_reset():
    c = _scene(20_000, seed=1)
    t = kdtree.New(c["base"])
    s = icp.IcpSession(t, c["target"], c["max_dist"], 6, None, c["threshold"], 5, BaseNormals=c["normals"])
    outs = []
    for rep in range(2):
        s.reset()
        for _ in range(5):
            s.step()
        tr, st, conv = s.result()
        outs.append(tr.copy())
        assert conv and st.NumIteration == 5
    assert np.array_equal(outs[0], outs[1])  # bitwise reproducible
    # partials -> (exchange) -> update is the same computation as the fused step
    s.reset()
    for _ in range(5):
        s.partials()
        s.update()
    tr2, _, _ = s.result()
    assert np.array_equal(tr2, outs[0])
    s.close()


def test_plane_singular_and_not_enough_pairs():
    # a flat base with parallel normals: x / y translation and rotation about z are unobservable
    rng = np.random.default_rng(0)
    base = np.zeros((5000, 3), f32)
    base[:, :2] = rng.random((5000, 2), dtype=f32) * f32(10)
    normals = np.tile(np.array([0, 0, 1], f32), (5000, 1))
    target = base + np.array([0.01, 0.0, 0.05], f32)
    t = kdtree.New(base)
    reg = icp.PointToPlaneICP(icp.PointToPlaneEvaluator(icp.NearestPointCorresponder(MaxDist=0.5), normals),
                              icp.GaussNewtonUpdaterFactory(Threshold=np.full(6, -1, f32), MaxIteration=3))
    with pytest.raises(icp.ErrSingular):
        reg.Fit(t, target)
    with pytest.raises(icp.ErrNotEnoughPairs):
        icp.PointToPlaneICP(icp.PointToPlaneEvaluator(icp.NearestPointCorresponder(MaxDist=0.001), normals,
                                                      MinPairs=6)).Fit(t, base[:3] + f32(5.0))


def test_plane_sharded_rccl_single_rank():
    """The 30-double exchange path (partials -> all-reduce -> update) through a 1-rank RCCL group."""
    import os
    import torch
    import torch.distributed as dist
    from pcgol_amd.distributed import ShardedIcp
    c = _scene(30_000, seed=2)
    torch.cuda.set_device(0)
    created = False
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        created = True
    try:
        t = kdtree.New(c["base"])
        a = ShardedIcp(t, c["target"], c["max_dist"], 6, None, c["threshold"], 6, force_exchange=True,
                       BaseNormals=c["normals"])
        assert a.exchange and a.sums.numel() == 30
        tr_a, st_a, _ = a.fit()
        b = ShardedIcp(t, c["target"], c["max_dist"], 6, None, c["threshold"], 6, BaseNormals=c["normals"])
        tr_b, st_b, _ = b.fit()
        assert np.array_equal(tr_a, tr_b) and st_a.NumIteration == st_b.NumIteration == 6
        a.close()
        b.close()
    finally:
        if created:
            dist.destroy_process_group()


def test_plane_c4_full_size():
    """BASELINE.json's fourth config as it is worded: ICP point-to-plane, 1M source vs 1M target, 20
    iterations (the extension: no reference to compare with).  The 30 sums of iteration 0 against the CPU
    oracle's float64 restatement; the pose after 20 Gauss-Newton steps against the synthetic ground truth;
    two runs bit-identical (fixed-order float64 reduction)."""
    c = synth.c4_plane(1_000_000)
    t = kdtree.New(c["base"])
    ev = icp.PointToPlaneEvaluator(icp.NearestPointCorresponder(MaxDist=c["max_dist"]), c["normals"], MinPairs=6)
    sums = ev.Sums(t, c["target"])
    osums = O.plane_sums(O.KDTree(c["base"]), c["normals"], c["target"], c["max_dist"])
    assert sums[29] == osums[29] and sums[28] == osums[28] and sums[29] > 0.99 * len(c["target"])
    scale = np.maximum(np.abs(osums), 1e-30)
    assert np.all(np.abs(sums - osums) <= 1e-11 * np.maximum(scale, osums[29]))
    reg = icp.PointToPlaneICP(ev, icp.GaussNewtonUpdaterFactory(Threshold=c["threshold"], MaxIteration=c["max_iteration"]))
    trans, stat = reg.Fit(t, c["target"])
    trans2, stat2 = reg.Fit(t, c["target"])
    assert stat.NumIteration == stat2.NumIteration == 20
    assert np.array_equal(trans, trans2) and np.array_equal(stat.Evaluated.Hessian, stat2.Evaluated.Hessian)
    inv = np.linalg.inv(synth.icp_pose().astype(np.float64).reshape(4, 4).T).T.reshape(-1)
    assert np.max(np.abs(trans.astype(np.float64) - inv)) <= 2e-6
    assert float(stat.Evaluated.Value) < 1e-9
