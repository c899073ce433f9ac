"""Counters of the instrumented walk: python tools/walk_stats.py {c2|c4|c4h} [presort] [iteration]
c4h = C4 at ICP iteration k >= 1 (default 1): queries re-projected by the pose after k updates,
pruning hints = the points matched at iteration k-1, as icp_corr_kernel sees them."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pcgol_amd import synth, kdtree, icp, _lib as L
which = sys.argv[1] if len(sys.argv) > 1 else "c4"
presort = int(sys.argv[2]) if len(sys.argv) > 2 else 1
L.check(L.lib().pcgx_init(0))
if which == "c2":
    c = synth.c2_knn(); base, q, mr = c["base"], c["queries"], 10.0
else:
    c = synth.c4_icp(); base, q, mr = c["base"], c["target"], 0.5
t = kdtree.New(base)
hint_ptr = None
if which == "c4h":
    k = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    s = icp.IcpSession(t, q, c["max_dist"], c["min_pairs"], c["weight"], c["threshold"], c["max_iteration"])
    poses = []
    for _ in range(k + 1):
        poses.append(s.result()[0].copy())
        s.step()
    s.close()
    q_prev = synth.transform_points(poses[k - 1], q) if k > 1 else q
    ids, _ = t.NearestBatch(q_prev, mr)
    hint = torch.from_numpy(np.ascontiguousarray(base[np.maximum(ids, 0)])).cuda()
    hint_ptr = L.ptr(hint.data_ptr())
    q = synth.transform_points(poses[k], q)
dq = torch.from_numpy(np.ascontiguousarray(q, dtype=np.float32)).cuda()
st = np.zeros(16, np.uint64)
L.check(L.lib().pcgx_debug_walk_stats(t._h, L.ptr(dq.data_ptr()), len(q), mr, presort, hint_ptr, L.ptr(st)))
it, act, look, refill, prep, ver, pend, nq, desc, epop, epass, ipop, ipass, leaf = [int(x) for x in st[:14]]
print("%s presort=%d: queries %d, iterations(wave) %d = %.3f/query, active lane-steps %.2f/query (util %.1f%%), "
      "node fetches %.2f/query, refill sections %d, chunks %d, verified %.1f%%, pend levels %.2f/query" %
      (which, presort, nq, it, it / nq, act / nq, 100.0 * act / (64.0 * it), look / nq, refill, prep,
       100.0 * ver / nq, pend / nq))
print("  per query: descending fetches %.2f (leaves %.2f), explicit pops %.2f (passing %.2f), first-descent pops %.2f (passing %.2f)"
      % (desc / nq, leaf / nq, epop / nq, epass / nq, ipop / nq, ipass / nq))
