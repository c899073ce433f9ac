"""CPU checks of the drop-in boundary: libpcgx.so loads without a GPU, exports every symbol
include/pcgx.h declares, fails loudly (no CPU fallback) when no device is present, and its
host-only functions agree with the oracle."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import oracle as O
from pcgol_amd import _lib as L
from pcgol_amd import icp, mat

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "pcgx.h")).read()
    return sorted(set(re.findall(r"PCGX_API\s+[\w\s\*]+?\b(pcgx_\w+)\s*\(", txt)))


def test_header_symbols_exported_and_bound():
    names = _declared()
    assert len(names) >= 35
    lib = C.CDLL(L.SO)
    for n in names:
        assert hasattr(lib, n), "libpcgx.so does not export %s" % n
    assert sorted(L.SIGNATURES) == names, "pcgol_amd/_lib.py binding list differs from include/pcgx.h"
    assert b"gfx950" in L.lib().pcgx_version()


def _has_gpu():
    import torch
    return torch.cuda.device_count() > 0


def test_no_gpu_fails_loudly():
    if _has_gpu():
        pytest.skip("a GPU is present")
    rc = L.lib().pcgx_init(0)
    assert rc == L.PCGX_E_HIP
    assert "no CPU fallback" in L.last_error()
    from pcgol_amd import kdtree
    with pytest.raises(L.PcgxError):
        kdtree.New(np.zeros((4, 3), np.float32))


def test_host_math_matches_oracle():
    rng = np.random.default_rng(0)
    for _ in range(200):
        v = (rng.random(3, dtype=np.float32) - np.float32(0.5)) * np.float32(rng.choice([0.05, 0.5, 3.0]))
        assert np.array_equal(mat.RodriguesToRotation(v).view(np.uint32), O.rodrigues(v).view(np.uint32))
    a = rng.random(16, dtype=np.float32)
    b = rng.random(16, dtype=np.float32)
    assert np.array_equal(mat.Mul(a, b), O.mat4_mul(a, b))
    pts = rng.random((100, 3), dtype=np.float32)
    m = O.mat4_mul(O.translate(1, 2, 3), O.rotate(0, 1, 0, 0.3))
    assert np.array_equal(mat.Transform(m, pts), O.mat4_transform(m, pts))
    assert np.array_equal(mat.Translate(4, 5, 6), O.translate(4, 5, 6))


def test_finish_evaluate_and_update_match_oracle():
    """Product host restatement (pcgx_math.h) vs oracle on the evaluator tail + updater."""
    rng = np.random.default_rng(1)
    base = rng.random((500, 3), dtype=np.float32) * np.float32(2)
    tgt = base[rng.permutation(500)[:300]] + np.float32(0.01)
    t = O.KDTree(base)
    o = O.icp_evaluate(t, tgt, 0.5, 6, sums_mode=1)
    ev = icp.FinishEvaluate(o["raw10"], 6)
    assert ev.Value == o["value"] and ev.DistRMS == o["dist_rms"] and ev.NumPairs == o["npairs"]
    assert np.array_equal(ev.Gradient, o["gradient"])
    with pytest.raises(icp.ErrNotEnoughPairs):
        icp.FinishEvaluate(o["raw10"], 301)
    u = icp.GradientDescentUpdaterFactory().New()
    tr = mat.Translate(0, 0, 0)
    otr, it = O.translate(0, 0, 0), 0
    for _ in range(25):
        tr, conv = u.Update(tr, ev)
        otr, oconv, it = O.icp_update(otr, o["gradient"], it)
        assert conv == oconv and np.array_equal(tr, otr) and u.i == it
        if conv:
            break
    assert u.i == 20  # default MaxIteration (updater.go:31-33)


def test_plane_finish_and_gauss_newton_match_oracle():
    """Point-to-plane extension (no reference parity): the product's host pieces (pcgx_math.h)
    against the oracle's independent restatement (oracle/plane_oracle.c), no GPU needed."""
    from pcgol_amd import synth
    c = synth.c4_plane(3000)
    t = O.KDTree(c["base"])
    sums = O.plane_sums(t, c["normals"], c["target"], c["max_dist"])
    oe = O.plane_finish(sums, 6)
    ev = icp.FinishEvaluatePlane(sums, 6)
    assert ev.Value == oe["value"] and ev.NumPairs == oe["npairs"]
    assert np.array_equal(ev.Gradient, oe["gradient"]) and np.array_equal(ev.Hessian, oe["hessian"])
    with pytest.raises(icp.ErrNotEnoughPairs):
        icp.FinishEvaluatePlane(sums, 3001)
    th = np.full(6, -1, np.float32)
    u = icp.GaussNewtonUpdaterFactory(Threshold=th, MaxIteration=4, Damping=0.01).New()
    tr, otr, it = mat.Translate(0, 0, 0), O.translate(0, 0, 0), 0
    for k in range(4):
        tr, conv = u.Update(tr, ev)
        otr, oconv, it = O.gauss_newton_update(otr, oe["gradient"], oe["hessian"], it, th, 0.01, 4)
        assert conv == oconv == (k == 3) and u.i == it
        assert np.max(np.abs(tr - otr)) <= 1e-7
    # flat test first (updater.go:45-54): a gradient inside the default thresholds converges at once
    small = icp.Evaluated()
    small.Gradient = np.full(6, 0.001, np.float32)
    small.Hessian = np.eye(6, dtype=np.float32).reshape(-1)
    assert icp.GaussNewtonUpdaterFactory().New().Update(mat.Translate(0, 0, 0), small)[1] is True
    # singular normal equations are an error, not a silent step
    bad = icp.Evaluated()
    bad.Gradient = np.ones(6, np.float32)
    bad.Hessian = np.zeros(36, np.float32)
    with pytest.raises(icp.ErrSingular):
        icp.GaussNewtonUpdaterFactory().New().Update(mat.Translate(0, 0, 0), bad)
    with pytest.raises(O.OracleError):
        O.gauss_newton_update(O.translate(0, 0, 0), bad.Gradient, bad.Hessian, 0)


def test_oracle_plane_fit_recovers_pose():
    """Sanity of the extension's oracle itself against synthetic ground truth."""
    from pcgol_amd import synth
    c = synth.c4_plane(4000)
    o = O.plane_fit(O.KDTree(c["base"]), c["normals"], c["target"], c["max_dist"], 6, c["threshold"], 0.0, 8)
    inv = np.linalg.inv(synth.icp_pose().astype(np.float64).reshape(4, 4).T).T.reshape(-1)
    assert o["num_iteration"] == 8
    assert np.max(np.abs(o["trans"].astype(np.float64) - inv)) < 5e-4
