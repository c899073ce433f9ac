"""One process, several device slots (pcgx_init_devices / pcgx_set_device / pcgx_icp_fit_multi; SURVEY 8(b) threading
row: "one process drives all 8 GPUs").  On the one-GPU test box the slots all name device 0: three independent sets of
streams and workspaces, a tree replica per slot, a host thread per slot inside the library, the exchange in host memory.

* the default sums (the reference's, over the slots' tiles one after the other) == the Go-semantics oracle's Fit on the
  whole target, bit for bit; the float64 mode == the one-GPU float64 Fit to rounding;
* a slot whose step fails in the middle of the Fit (fault injection) makes EVERY slot return within the timeout --
  nobody is left inside an all-reduce (ADVICE round 2 / VERDICT round 3).

The ring form of the reference sums (csrc/strict.hip, strict_enqueue_ring) lets every slot's kernels wait for the slot
before it ON THE DEVICE.  HIP hands its few hardware queues to streams as they have work, and a kernel queued behind
the one that waits for it never starts: pcgx_init_devices therefore gives the library's stream of every slot a
hardware queue of its own when slots share a device (a stream made with a CU mask; csrc/core.hip) -- what separate
GPUs have by themselves.  (Found by running these tests in fresh processes: eight slots stood still until a wait ran
out of time.)"""
import ctypes as C
import time

import numpy as np
import pytest

import oracle as O
from pcgol_amd import _lib as L
from pcgol_amd import icp, kdtree, synth

pytestmark = pytest.mark.gpu
N_SLOTS = 8   # the target machine's GPU count; every test below runs with 3 and with 8 of them


@pytest.fixture(scope="module")
def slots():
    ids = np.zeros(N_SLOTS, np.int32)
    L.check(L.lib().pcgx_init_devices(N_SLOTS, L.ptr(ids)))
    yield N_SLOTS
    L.check(L.lib().pcgx_set_device(0))


def _fit_multi(c, trees, tiles, sums_mode=0):
    n = len(trees)
    params = icp._params(c["max_dist"], 0.0, c["min_pairs"], c["weight"], c["threshold"], c["max_iteration"], sums_mode=sums_mode)
    bases = (C.c_void_p * n)(*[t._h for t in trees])
    tps = (C.c_void_p * n)(*[t.ctypes.data for t in tiles])
    nts = (C.c_int64 * n)(*[len(t) for t in tiles])
    trans = np.empty(16, np.float32)
    st = L.IcpStat()
    rc = L.lib().pcgx_icp_fit_multi(n, bases, tps, nts, C.byref(params), L.ptr(trans), C.byref(st))
    return rc, trans, st


def _trees(base, n):
    trees = []
    for r in range(n):
        L.check(L.lib().pcgx_set_device(r))
        trees.append(kdtree.New(base))     # the replica of slot r
    L.check(L.lib().pcgx_set_device(0))
    return trees


@pytest.mark.parametrize("ring", ["1", "0"], ids=["ring", "collectives"])
@pytest.mark.parametrize("ns", [3, 8])
def test_fit_multi_reference_sums_equal_the_oracle(slots, ns, ring, monkeypatch):
    """ring: every slot's kernels resident at once, the walk handed from slot to slot through host-coherent words
    (csrc/strict.hip, strict_enqueue_ring); collectives (PCGX_SHARD_RING=0): the 2 + world all-reduces it replaces."""
    monkeypatch.setenv("PCGX_SHARD_RING", ring)
    slots = ns
    n = 200_000
    c = synth.c4_icp(n=n, width=10.0 * (n / 1e6) ** (1 / 3))
    trees = _trees(c["base"], slots)
    cuts = [0, 70_001, 130_000, n] if ns == 3 else [0, 70_001, 70_002, 70_002, 130_000, 131_000, 150_000, 199_999, n]   # ragged, not tile aligned
    tiles = [np.ascontiguousarray(c["target"][cuts[r]:cuts[r + 1]]) for r in range(slots)]
    stats = np.zeros(4, np.int64)
    L.check(L.lib().pcgx_debug_shard_stats(L.ptr(stats), 1))
    kinds = np.zeros(2, np.int64)
    L.check(L.lib().pcgx_debug_ring_kinds(L.ptr(kinds), 1))
    rc, trans, st = _fit_multi(c, trees, tiles)
    L.check(rc)
    L.check(L.lib().pcgx_debug_shard_stats(L.ptr(stats), 1))
    L.check(L.lib().pcgx_debug_ring_kinds(L.ptr(kinds), 1))
    # the slots' inboxes: in device memory, one another's plain pointers (one process)
    assert kinds.tolist() == ([1, 0] if ring == "1" else [0, 0]), kinds
    # every slot enqueued its 20 steps in the form asked for (one ring per Fit, in the process's pinned memory)
    assert (stats[0], stats[1], stats[2]) == ((20 * ns, 0, 1) if ring == "1" else (0, 20 * ns, 0)), stats
    o32 = O.icp_fit(O.KDTree(c["base"]), c["target"], c["max_dist"], c["min_pairs"], c["weight"], c["threshold"],
                    c["max_iteration"], sums_mode=0)
    assert st.num_iteration == o32["num_iteration"] == 20
    assert np.array_equal(trans.ravel(), np.asarray(o32["trans"]).ravel())
    assert np.float32(st.evaluated.value) == o32["value"]
    assert np.array_equal(np.array(st.evaluated.gradient, np.float32), o32["gradient"])
    # the float64 mode: one all-reduce per iteration, the one-GPU float64 Fit to rounding
    rc, t64, st64 = _fit_multi(c, trees, tiles, sums_mode=icp.SumsF64Tree)
    L.check(rc)
    reg = icp.PointToPointICPGradient(
        icp.PointToPointEvaluator(icp.NearestPointCorresponder(MaxDist=c["max_dist"]), MinPairs=c["min_pairs"],
                                  SumsMode=icp.SumsF64Tree),
        icp.GradientDescentUpdaterFactory(Weight=c["weight"], Threshold=c["threshold"], MaxIteration=c["max_iteration"]))
    t1, s1 = reg.Fit(trees[0], c["target"])
    assert st64.num_iteration == s1.NumIteration == 20 and np.max(np.abs(t64.ravel() - np.asarray(t1).ravel())) <= 1e-6


@pytest.mark.parametrize("ns", [3, 8])
@pytest.mark.parametrize("mode", [0, 1, 2], ids=["reference", "f64", "reference-collectives"])
def test_a_failing_slot_ends_the_fit_on_every_slot(slots, mode, ns, monkeypatch):
    slots = ns
    monkeypatch.setenv("PCGX_SHARD_RING", "0" if mode == 2 else "1")
    n = 60_000
    c = synth.c4_icp(n=n, width=10.0 * (n / 1e6) ** (1 / 3))
    trees = _trees(c["base"], slots)
    tiles = [np.ascontiguousarray(c["target"][r::slots]) for r in range(slots)]
    monkeypatch.setenv("PCGX_TEST_FAIL_RANK", "1")
    monkeypatch.setenv("PCGX_TEST_FAIL_ITER", "7")
    t0 = time.time()
    rc, trans, st = _fit_multi(c, trees, tiles, sums_mode=icp.SumsF64Tree if mode == 1 else 0)
    assert time.time() - t0 < 30.0
    assert rc != 0
    buf = C.create_string_buffer(512)
    L.lib().pcgx_last_error(buf, 512)
    assert b"injected failure of rank 1 in iteration 7" in buf.value, buf.value
    monkeypatch.delenv("PCGX_TEST_FAIL_RANK")
    rc, trans, st = _fit_multi(c, trees, tiles, sums_mode=icp.SumsF64Tree if mode == 1 else 0)   # and the library still works
    L.check(rc)
    assert st.num_iteration == 20


@pytest.mark.parametrize("ns", [3, 8])
def test_fit_multi_edge_shards(slots, ns, monkeypatch):
    """The reference's sums over slots whose shards are awkward: one slot with NO targets at all, one with a single
    target, targets that find no partner scattered through the order, a built-in weight (nine chained sums: the sum of
    the weights goes round the slots like the others) -- Evaluated and pose of the oracle's Fit on the concatenated
    target, bit for bit."""
    monkeypatch.setenv("PCGX_SHARD_RING", "1")
    n = 90_000
    c = synth.c4_icp(n=n, width=10.0 * (n / 1e6) ** (1 / 3))
    target = c["target"].copy()
    far = target[::41] + np.float32(50.0)
    target = np.ascontiguousarray(np.insert(target, np.arange(0, len(far)) * 33, far, axis=0))
    trees = _trees(c["base"], ns)
    if ns == 3:
        tiles = [np.ascontiguousarray(target[:0]), np.ascontiguousarray(target[:1]), np.ascontiguousarray(target[1:])]
    else:   # empty slots at either end and in the middle, a single target, a slot that holds less than one tile
        cuts = [0, 0, 1, 1, 1501, 40_000, 40_000, len(target), len(target)]
        tiles = [np.ascontiguousarray(target[cuts[r]:cuts[r + 1]]) for r in range(ns)]
    wf = icp.WeightHuber(0.0009)
    O.set_weight_fn(wf.kind, wf.a)
    try:
        n_s = len(trees)
        params = icp._params(0.05, 0.0, c["min_pairs"], c["weight"], c["threshold"], c["max_iteration"], weight_fn=wf)
        bases = (C.c_void_p * n_s)(*[t._h for t in trees])
        tps = (C.c_void_p * n_s)(*[t.ctypes.data for t in tiles])
        nts = (C.c_int64 * n_s)(*[len(t) for t in tiles])
        trans = np.empty(16, np.float32)
        st = L.IcpStat()
        L.check(L.lib().pcgx_icp_fit_multi(n_s, bases, tps, nts, C.byref(params), L.ptr(trans), C.byref(st)))
        o32 = O.icp_fit(O.KDTree(c["base"]), target, 0.05, c["min_pairs"], c["weight"], c["threshold"], c["max_iteration"],
                        sums_mode=0)
        assert st.num_iteration == o32["num_iteration"]
        assert np.array_equal(trans.ravel(), np.asarray(o32["trans"]).ravel())
        assert np.float32(st.evaluated.value) == o32["value"]
        assert np.array_equal(np.array(st.evaluated.gradient, np.float32), o32["gradient"])
        assert 6 <= st.evaluated.num_pairs < len(target)
    finally:
        O.set_weight_fn(0, 0.0)
