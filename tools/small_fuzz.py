"""A soak of the one-launch Fit (csrc/icp_small.hip) against the oracle: random sizes, clouds (a surface, a lattice, a plane,
twins), MaxDist / MinDistSq, weights, ways through the tree (PCGX_ICP_SMALL_HIER / _P / _ORDER_FROM are read at every launch).
    python tools/small_fuzz.py [cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle as O
from pcgol_amd import _lib as L
from pcgol_amd import icp, kdtree, synth

f32 = np.float32
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
os.environ["PCGX_ICP_SMALL_BASE"], os.environ["PCGX_ICP_SMALL_TARGET"] = "65535", "16384"
bad = 0
for case in range(cases):
    nb = int(rng.choice([1, 2, 3, 7, 64, 65, 500, 1000, 2047, 2048, 2049, 4096, 5000, 9000, 16383, 20000]))
    nt = int(rng.choice([1, 2, 63, 64, 65, 300, 1000, 2048, 2049, 4097, 6000, 12000, 16384]))
    if nb * nt > 60_000_000:
        nt = max(1, 60_000_000 // nb)
    kind = int(rng.integers(0, 4))
    if kind == 0:
        c = synth.c4_icp(n=max(nb, nt), width=float(rng.uniform(1.0, 4.0)))
        base, target = np.ascontiguousarray(c["base"][:nb]), np.ascontiguousarray(c["target"][:nt])
        max_dist = 0.5
    elif kind == 1:
        base = rng.integers(0, 12, (nb, 3)).astype(f32)
        target = (rng.integers(0, 24, (nt, 3)) * 0.5 + rng.uniform(-0.05, 0.05, 3)).astype(f32)
        max_dist = float(rng.choice([0.5, 1.0, 3.0]))
    elif kind == 2:
        base = np.concatenate([rng.uniform(-5, 5, (nb, 2)), np.zeros((nb, 1))], axis=1).astype(f32)
        target = (base[rng.integers(0, nb, nt)] + rng.normal(0, 0.05, (nt, 3)) + np.array([0.1, -0.05, 0.02])).astype(f32)
        max_dist = 1.0
    else:
        half = rng.uniform(-2, 2, (nb // 2 + 1, 3)).astype(f32)
        base = np.concatenate([half, half])[:nb]
        target = (base[rng.integers(0, nb, nt)] + rng.normal(0, 0.02, (nt, 3))).astype(f32)
        max_dist = 0.7
    mds = float(rng.choice([0.0, 0.0, 0.0025, 0.04]))
    iters = int(rng.choice([1, 3, 8]))
    w, th = np.full(6, float(rng.choice([0.05, 0.3])), f32), np.full(6, -1.0, f32)
    min_pairs = int(rng.choice([1, 6]))
    os.environ["PCGX_ICP_SMALL_HIER"] = str(int(rng.choice([-1, 0, 1, 2, 3])))
    p = int(rng.choice([0, 0, 1, 2, 3, 5]))
    if p and p * ((nt + 63) // 64) <= 256:
        os.environ["PCGX_ICP_SMALL_P"] = str(p)
    else:
        os.environ.pop("PCGX_ICP_SMALL_P", None)
    os.environ["PCGX_ICP_SMALL_ORDER_FROM"] = str(int(rng.choice([0, 2048, 100000])))
    o = None
    try:
        o = O.icp_fit(O.KDTree(base, mds), target, max_dist, min_pairs, w, th, iters, sums_mode=0)
    except O.OracleError:
        pass
    reg = icp.PointToPointICPGradient(icp.PointToPointEvaluator(icp.NearestPointCorresponder(MaxDist=max_dist), MinPairs=min_pairs),
                                      icp.GradientDescentUpdaterFactory(Weight=w, Threshold=th, MaxIteration=iters))
    ok = True
    try:
        trans, st = reg.Fit(kdtree.New(base, MinDistSq=mds), target)
        ok = o is not None and st.NumIteration == o["num_iteration"] and np.array_equal(np.asarray(trans, f32).ravel(), np.asarray(o["trans"], f32).ravel()) \
            and f32(st.Evaluated.Value) == o["value"]
    except L.PcgxError:
        ok = o is None
    if not ok:
        bad += 1
        print("MISMATCH case %d: nb %d nt %d kind %d max_dist %g mds %g iters %d env %s %s %s" % (
            case, nb, nt, kind, max_dist, mds, iters, os.environ["PCGX_ICP_SMALL_HIER"], os.environ.get("PCGX_ICP_SMALL_P"), os.environ["PCGX_ICP_SMALL_ORDER_FROM"]), flush=True)
    if case % 25 == 24:
        print("... %d cases, %d mismatches" % (case + 1, bad), flush=True)
out = (__import__("ctypes").c_int64 * 3)()
L.check(L.lib().pcgx_debug_icp_one_launch(out, 0))
print("%d cases, %d mismatches; %d one-launch Fits (%d below what could not be ruled out only, %d with the targets grouped)" % (cases, bad, out[0], out[1], out[2]))
sys.exit(1 if bad else 0)
