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
leaf_ptr = None
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
    # first-descent leaves of iteration k-1 (an instrumented run over the previous positions)
    leaves = torch.zeros(len(q), dtype=torch.int32).cuda()
    leaf_ptr = L.ptr(leaves.data_ptr())
    dqp = torch.from_numpy(np.ascontiguousarray(q_prev, dtype=np.float32)).cuda()
    L.check(L.lib().pcgx_debug_walk_stats(t._h, L.ptr(dqp.data_ptr()), len(q), mr, presort, None, leaf_ptr,
                                          L.ptr(np.zeros(32, np.uint64))))
    q = synth.transform_points(poses[k], q)
dq = torch.from_numpy(np.ascontiguousarray(q, dtype=np.float32)).cuda()
st = np.zeros(32, np.uint64)
L.check(L.lib().pcgx_debug_walk_stats(t._h, L.ptr(dq.data_ptr()), len(q), mr, presort, hint_ptr, leaf_ptr, L.ptr(st)))
it, act, look, refill, prep, ver, pend, nq, desc, epop, epass, ipop, ipass, leaf = [int(x) for x in st[:14]]
tail, itmax = int(st[14]), int(st[15])
early, tsteps, titer = int(st[21]), int(st[22]), int(st[23])
print("%s presort=%d: queries %d, iterations(wave) %d = %.3f/query, active lane-steps %.2f/query (util %.1f%%), "
      "node fetches %.2f/query, refill sections %d, chunks %d, verified %.1f%%, pend levels %.2f/query" %
      (which, presort, nq, it, it / nq, act / nq, 100.0 * act / (64.0 * it), look / nq, refill, prep,
       100.0 * ver / nq, pend / nq))
print("  per query: descending fetches %.2f (leaves %.2f), explicit pops %.2f (passing %.2f), first-descent pops %.2f (passing %.2f)"
      % (desc / nq, leaf / nq, epop / nq, epass / nq, ipop / nq, ipass / nq))
print("  finished inside the preparation %.1f%%; its descent loop: %.2f lane-steps/query, %.2f iterations/chunk"
      % (100.0 * early / nq, tsteps / nq, titer / max(prep, 1)))
print("  iterations after a wave's last hand-out %.1f%% of all, most iterations of one wave %d" % (100.0 * tail / it, itmax))
w = max(int(st[19]), 1)
print("  per wave us: query fetch %.1f, path fetch+verify %.1f, descent loop %.1f, leaf+emit+enqueue %.1f, take %.1f, stepping %.1f"
      % tuple(int(st[k]) / w / 100.0 for k in range(24, 30)))
print("  instrumented kernel %.1f us; per wave: %.0f ticks until the last hand-out, %.0f after it; longest wave %d ticks"
      % (int(st[20]) / 1e3, int(st[16]) / w, int(st[17]) / w, int(st[18])))
if os.environ.get("PCGX_DEBUG_WALK_ROWS"):
    r = np.fromfile(os.environ["PCGX_DEBUG_WALK_ROWS"], np.uint64).reshape(-1, 32).astype(np.float64)
    dur, it_w, dry = r[:, 18] / 100.0, r[:, 0], r[:, 16] / 100.0
    pct = lambda a: " ".join("%.1f" % x for x in np.percentile(a, [0, 10, 50, 90, 99, 100]))
    print("  per wave (min p10 p50 p90 p99 max): duration us %s | until last hand-out us %s | iterations %s | queries %s"
          % (pct(dur), pct(dry), pct(it_w), pct(r[:, 7])))
    blk = dur.reshape(-1, 4).max(axis=1)
    print("  per block duration us: %s; per XCD (block %% 8) mean of block durations: %s"
          % (pct(blk), " ".join("%.1f" % blk[i::8].mean() for i in range(8))))
