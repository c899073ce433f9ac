"""BASELINE config C5 (ICP on a 64M-point cloud tiled over 8 GPUs), the share ONE rank holds: the
replicated 64M-point base tree and its ~8M-point target tile (synth.c5_tile: octant 0).

Checkers at this size: brute force on the device (torch elementwise float32 ops in the reference's
expression order, (dx*dx + dy*dy) + dz*dz, no fusion) and the product's own tree walk in the
reference's visit order (PCGX_GRID=0) against its certified grid pass; and the CPU oracle itself
through committed digests (tests/golden/c5_digest.json, written by tests/golden/make_c5_digest.py:
the oracle's O(N log^2 N) tree build takes ~9 minutes at 64M points, so it runs once, not per test run)."""
import numpy as np
import pytest

from pcgol_amd import _lib as L
from pcgol_amd import icp, kdtree, synth

pytestmark = pytest.mark.gpu

NB = 64_000_000
WIDTH = 40.0


@pytest.fixture(scope="module")
def c5():
    base = synth.uniform_cloud_chunked(NB, WIDTH, 2)
    tile = synth.c5_tile(base, 0, 8, WIDTH)
    tree = kdtree.New(base)
    yield base, tile, tree
    del tree


def brute_force(base, q):
    import torch
    dq = torch.from_numpy(q).cuda()
    best_d = torch.full((len(q),), float("inf"), device="cuda")
    best_i = torch.full((len(q),), -1, dtype=torch.int64, device="cuda")
    chunk = 1_000_000
    for s in range(0, len(base), chunk):
        b = torch.from_numpy(base[s:s + chunk]).cuda()
        best = None
        for c0 in range(0, len(q), 256):
            d = b[None, :, :] - dq[c0:c0 + 256, None, :]
            d2 = d * d
            dist = (d2[..., 0] + d2[..., 1]) + d2[..., 2]
            m, i = dist.min(dim=1)
            upd = m < best_d[c0:c0 + 256]   # strict: the lowest index among equal minima stays
            best_d[c0:c0 + 256] = torch.where(upd, m, best_d[c0:c0 + 256])
            best_i[c0:c0 + 256] = torch.where(upd, i + s, best_i[c0:c0 + 256])
    return best_i.cpu().numpy(), best_d.cpu().numpy()


def test_c5_tile_is_an_eighth_and_tree_depth(c5):
    base, tile, tree = c5
    assert abs(len(tile) - NB // 8) < NB // 8 * 0.01
    assert tree.Len() == NB and tree.MaxDepth() == 26


def test_c5_nearest_equals_brute_force(c5):
    """1024 targets of the tile against all 64M base points: ids and DistSq bit for bit (no exact ties
    in this cloud: the brute force's lowest-index rule and the tree's visit order agree)."""
    base, tile, tree = c5
    q = np.ascontiguousarray(tile[:: len(tile) // 1024][:1024])
    ids, dsq = tree.NearestBatch(q, 0.5)
    bi, bd = brute_force(base, q)
    inr = bd <= np.float32(0.5) * np.float32(0.5)
    assert inr.sum() > 1000
    assert np.array_equal(dsq[inr], bd[inr])
    assert np.array_equal(ids[inr], bi[inr])
    assert np.all(ids[~inr] == -1)


def test_c5_pairs_grid_pass_equals_reference_order_walk(c5, monkeypatch):
    """Iteration 0's correspondences of a 1M-target slice of the tile: the certified grid pass against
    the tree walk in the reference's visit order (kdtree.go:94-146), pair for pair."""
    base, tile, tree = c5
    q = np.ascontiguousarray(tile[:1_000_000])
    ids_g, dsq_g = tree.NearestBatch(q, 0.5)
    monkeypatch.setenv("PCGX_GRID", "0")
    ids_w, dsq_w = tree.NearestBatch(q, 0.5)
    assert np.array_equal(ids_g, ids_w) and np.array_equal(dsq_g, dsq_w)
    assert (ids_g >= 0).mean() > 0.99


def test_c5_against_the_oracle_digest(c5):
    """What the CPU oracle computed on this configuration (tests/golden/c5_digest.json): Nearest of the
    first 100k targets of the tile against the 64M-point tree (ids and DistSq bits), iteration 0's pairs of
    the first 1M targets, and its evaluator sums -- the float64 tree's ten sums to rounding of the
    summation order, the reference's sequential float32 sums (the default mode) bit for bit."""
    import json
    import os
    base, tile, tree = c5
    with open(os.path.join(os.path.dirname(__file__), "golden", "c5_digest.json")) as f:
        g = json.load(f)
    assert g["n_base"] == len(base) and g["n_tile"] == len(tile)
    q = np.ascontiguousarray(tile[: g["nearest"]["n"]])
    ids, dsq = tree.NearestBatch(q, g["max_dist"])
    assert int((ids >= 0).sum()) == g["nearest"]["found"] and int(ids.sum()) == g["nearest"]["ids_sum"]
    assert int(np.bitwise_xor.reduce(ids)) == g["nearest"]["ids_xor"]
    assert int(np.bitwise_xor.reduce(dsq.view(np.uint32))) == g["nearest"]["dsq_bits_xor"]
    assert [int(v) for v in ids[:64]] == g["nearest"]["first_ids"]
    t1m = np.ascontiguousarray(tile[: g["pairs"]["n_target"]])
    b, t, d = icp.NearestPointCorresponder(MaxDist=g["max_dist"]).PairsArrays(tree, t1m)
    assert len(b) == g["pairs"]["n_pairs"] and int(np.bitwise_xor.reduce(b)) == g["pairs"]["base_ids_xor"]
    assert int(t.sum()) == g["pairs"]["target_ids_sum"]
    assert int(np.bitwise_xor.reduce(d.view(np.uint32))) == g["pairs"]["dsq_bits_xor"]
    # the sums of iteration 0 over those pairs
    s64 = icp.IcpSession(tree, t1m, g["max_dist"], 6, SumsMode=icp.SumsF64Tree)
    s64.partials()
    got = s64.read_sums()
    s64.close()
    want = np.array(g["sums"]["f64_tree_raw10"])
    assert got[9] == want[9] == g["pairs"]["n_pairs"]
    scale = np.maximum(np.abs(want), want[9] * 1e-3)
    assert np.all(np.abs(got - want) <= 1e-11 * scale), (got, want)
    ev = icp.PointToPointEvaluator(icp.NearestPointCorresponder(MaxDist=g["max_dist"]), MinPairs=6).Evaluate(tree, t1m)
    assert int(np.float32(ev.Value).view(np.uint32)) == g["sums"]["reference_value_bits"]
    assert [int(v) for v in ev.Gradient.view(np.uint32)] == g["sums"]["reference_gradient_bits"]
    assert int(np.float32(ev.DistRMS).view(np.uint32)) == g["sums"]["reference_dist_rms_bits"]


def test_c5_partial_sums_reproducible_over_a_fit(c5):
    """20 iterations on the tile, the rank's 10 partial sums after every correspondence pass: a
    second run gives the same bits (fixed-order float64 reduction), and the pose moves towards the
    inverse of the synthetic displacement."""
    base, tile, tree = c5
    cfg = dict(MaxDist=0.5, MinPairs=6, Weight=np.full(6, 0.3, np.float32), Threshold=np.full(6, -1.0, np.float32),
               MaxIteration=20)
    runs = []
    for rep in range(2):
        s = icp.IcpSession(tree, tile, SumsMode=icp.SumsF64Tree, **cfg)
        sums = []
        for _ in range(20):
            s.partials()
            sums.append(s.read_sums().copy())
            s.update()
        tr, st, conv = s.result()
        runs.append((np.array(sums), tr))
        s.close()
    assert np.array_equal(runs[0][0].view(np.uint64), runs[1][0].view(np.uint64))
    assert np.array_equal(runs[0][1], runs[1][1])
    assert runs[0][0][0][9] > 0.99 * len(tile)   # pairs
    inv = np.linalg.inv(synth.icp_pose().astype(np.float64).reshape(4, 4).T).T.reshape(-1)
    assert np.max(np.abs(runs[0][1].astype(np.float64) - inv)) < 0.02


def test_c5_reference_sums_on_the_whole_tile_against_the_oracle_digest(c5):
    """The reference's sums over the WHOLE 8M-target tile (3907 tiles per sum, eight chunk workgroups of the chain
    kernel per sum) against what the CPU oracle computed there (tests/golden/c5_digest.json "tile": its sequential
    float32 sums): iteration 0's Evaluated, and pose + Evaluated of a three-iteration Fit, bit for bit."""
    import json
    import os
    base, tile, tree = c5
    with open(os.path.join(os.path.dirname(__file__), "golden", "c5_digest.json")) as f:
        g = json.load(f)["tile"]
    ev = icp.PointToPointEvaluator(icp.NearestPointCorresponder(MaxDist=0.5), MinPairs=6).Evaluate(tree, tile)
    assert ev.NumPairs == g["n_pairs"]
    assert int(np.float32(ev.Value).view(np.uint32)) == g["evaluate_value_bits"]
    assert [int(v) for v in ev.Gradient.view(np.uint32)] == g["evaluate_gradient_bits"]
    assert int(np.float32(ev.DistRMS).view(np.uint32)) == g["evaluate_dist_rms_bits"]
    s = icp.IcpSession(tree, tile, MaxDist=0.5, MinPairs=6, Weight=np.full(6, 0.3, np.float32),
                       Threshold=np.full(6, -1.0, np.float32), MaxIteration=3)
    for _ in range(3):
        s.step()
    tr, st, _ = s.result()
    sst = s.strict_stats()
    s.close()
    assert st.NumIteration == g["fit3_num_iteration"] == 3
    assert [int(v) for v in np.asarray(tr, np.float32).ravel().view(np.uint32)] == g["fit3_trans_bits"]
    assert int(np.float32(st.Evaluated.Value).view(np.uint32)) == g["fit3_value_bits"]
    assert [int(v) for v in np.asarray(st.Evaluated.Gradient, np.float32).view(np.uint32)] == g["fit3_gradient_bits"]
    assert int(np.float32(st.Evaluated.DistRMS).view(np.uint32)) == g["fit3_dist_rms_bits"]
    assert sst[62] == 0 and sst[63] == 0   # no walker walked alone, no workgroup gave up the exchange


def test_c5_strict_sums_on_the_tile(c5, monkeypatch):
    """The parallel strict sums over 8M targets (3907 tiles per sum: eight chunks of the chain kernel,
    62 level-1 bins) against the one-wave chain, three iterations, with the in-kernel self-check (the oracle's own
    sums over this tile: the test above)."""
    monkeypatch.setenv("PCGX_STRICT_SELFCHECK", "1")
    base, tile, tree = c5
    cfg = dict(MaxDist=0.5, MinPairs=6, Weight=np.full(6, 0.3, np.float32), Threshold=np.full(6, -1.0, np.float32),
               MaxIteration=20)
    a = icp.IcpSession(tree, tile, **cfg)
    b = icp.IcpSession(tree, tile, **cfg)
    a.set_strict(1)
    b.set_strict(2)
    for k in range(3):
        a.step()
        b.step()
        sa, sb = a.read_sums(), b.read_sums()
        assert np.array_equal(sa.view(np.uint64), sb.view(np.uint64)), (k, sa, sb)
        st = a.strict_stats()
        assert not st[12:16].any() and st[6] == 0 and st[7] == 0, (k, st[:16])
    a.close()
    b.close()


def test_c5_walks_ahead_of_the_chunks_hand_overs(c5, monkeypatch):
    """Round 6: a chunk's walker that has to wait for its start state walks ahead of the wait, from the 256 states around
    the float64 guess (csrc/strict.hip, strict_chain_kernel<., kSpec>): at C5's eight chunks per sum the seven later
    chunks' walks run beside the first one's instead of behind it.  Counted (strict stats [10] walks carried through,
    [11] whose candidates held the state that came), and the sums are the one-wave chain's.  With the guess pushed off
    by 100 000 floats (PCGX_TEST_SPEC_MISS) every such walk MISSES, and the real walk runs behind it as it always
    did (the walk ahead leaves the helper waves and their tables where they are): the same bits."""
    base, tile, tree = c5
    cfg = dict(MaxDist=0.5, MinPairs=6, Weight=np.full(6, 0.3, np.float32), Threshold=np.full(6, -1.0, np.float32),
               MaxIteration=20)
    b = icp.IcpSession(tree, tile, **cfg)
    b.set_strict(2)
    ref = []
    for k in range(3):
        b.step()
        ref.append(b.read_sums().copy())
    b.close()
    for miss in (False, True):
        if miss:
            monkeypatch.setenv("PCGX_TEST_SPEC_MISS", "1")
        a = icp.IcpSession(tree, tile, **cfg)
        a.strict_stats()
        walked, hit = 0, 0
        for k in range(3):
            a.step()
            assert np.array_equal(a.read_sums().view(np.uint64), ref[k].view(np.uint64)), (miss, k)
            st = a.strict_stats()
            walked += int(st[10])
            hit += int(st[11])
            assert st[62] == 0
        a.close()
        if miss:
            assert walked > 0 and hit == 0, (walked, hit)
        else:
            assert walked > 0 and hit >= 0.5 * walked, (walked, hit)   # (measured: 85 % of the walks at this size)
    monkeypatch.delenv("PCGX_TEST_SPEC_MISS")
