"""The reference's Nearest (kdtree.go:83-146) WITHOUT a walk: every point's distance, and the walk's answer picked out of
them by the point's place in the query's visit order.  A model of csrc/icp_small.hip's search, checked against the oracle.

The walk is the in-order traversal  near sub-tree ; node ; far sub-tree  with one running best (knn_walk.h).  What it
skips (kdtree.go:111-115: plane distance^2 > best) holds only points STRICTLY farther than the best -- float32 sums of
squares are monotone -- so the answer is that of the full traversal:
  * MinDistSq cut (:104-106,120-122,140-142): the FIRST point in visit order with DistSq < MinDistSq, if there is one;
  * else the smallest DistSq d*; of the points at d* the first in visit order takes the best, a LEAF at d* behind it takes
    it again (:100 replaces unless strictly farther; a pivot needs strictly nearer, :117): the last leaf at d*, else the
    first point at d*;  d* > maxRange^2: nothing; d* == maxRange^2: leaves only.
Visit order as a number: two bits per level, 0 = in the near child's sub-tree, 1 = this node, 2 = in the far child's."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import oracle as O

f32 = np.float32


def implicit_tree(t):
    """BFS slots of the oracle's tree: slot 1 the root, children 2b / 2b + 1 (a node of two points has child0 only)."""
    d = t.dump()  # rows [id, dim, child0, child1] (row numbers)
    slots = {}
    def rec(k, b):
        if k < 0:
            return
        slots[b] = (int(d[k][0]), int(d[k][1]), int(d[k][2]) >= 0, int(d[k][3]) >= 0)
        rec(int(d[k][2]), 2 * b)
        rec(int(d[k][3]), 2 * b + 1)
    if len(d):
        rec(0, 1)
    return slots


def order_search(pts, slots, q, max_range, min_dist_sq):
    R = f32(max_range) * f32(max_range)
    D = max(b.bit_length() for b in slots)
    best_q = None   # (key, b)
    fd = None       # (dd, key, b)   first point at the smallest distance
    ld = None       # (dd, -key, b)  last LEAF at the smallest leaf distance
    q = np.asarray(q, f32)
    for b, (pid, dim, c0, c1) in slots.items():
        depth = b.bit_length() - 1
        key = 0
        for j in range(depth):
            a = b >> (depth - j)
            c = (b >> (depth - j - 1)) & 1
            aid, adim, a0, a1 = slots[a]
            assert adim == j % 3
            near1 = 0 if not (a0 and a1) else (0 if pts[aid][adim] > q[adim] else 1)
            if not (a0 and a1):
                assert a0 and not a1 and c == 0
            key |= (2 if c != near1 else 0) << (2 * (D - 1 - j))
        key |= 1 << (2 * (D - 1 - depth))
        dv = pts[pid] - q
        dd = f32(f32(dv[0] * dv[0] + dv[1] * dv[1]) + dv[2] * dv[2])
        leaf = not c0 and not c1
        if dd < f32(min_dist_sq) and (best_q is None or key < best_q[0]):
            best_q = (key, b)
        if fd is None or (dd, key) < (fd[0], fd[1]):
            fd = (dd, key, b)
        if leaf and (ld is None or (dd, -key) < (ld[0], ld[1])):
            ld = (dd, -key, b)
    if best_q is not None:
        b = best_q[1]
    else:
        dmin = fd[0]
        if dmin > R:
            return -1, R
        if ld is not None and ld[0] == dmin:
            b = ld[2]
        elif dmin == R:
            return -1, R
        else:
            b = fd[2]
    pid = slots[b][0]
    dv = pts[pid] - q
    return pid, f32(f32(dv[0] * dv[0] + dv[1] * dv[1]) + dv[2] * dv[2])


def check(pts, qs, max_range, mds):
    t = O.KDTree(pts, mds)
    slots = implicit_tree(t)
    ids, dsq = t.nearest_batch(qs, max_range)
    bad = 0
    for i, q in enumerate(qs):
        pid, dd = order_search(t.pts, slots, q, max_range, mds)
        if pid != ids[i] or f32(dd).tobytes() != f32(dsq[i]).tobytes():
            bad += 1
            if bad < 5:
                print("MISMATCH", i, q, (pid, dd), (ids[i], dsq[i]))
    return bad


if __name__ == "__main__":
    rng = np.random.default_rng(5)
    total = 0
    for trial in range(60):
        n = int(rng.integers(1, 200))
        kind = trial % 4
        if kind == 0:
            pts = rng.uniform(-1, 1, (n, 3)).astype(f32)
        elif kind == 1:   # a lattice: ties everywhere
            pts = rng.integers(0, 4, (n, 3)).astype(f32)
        elif kind == 2:   # a plane
            pts = np.concatenate([rng.integers(0, 6, (n, 2)).astype(f32), np.zeros((n, 1), f32)], axis=1)
        else:             # duplicates
            pts = rng.integers(0, 2, (n, 3)).astype(f32)
        qs = np.concatenate([rng.uniform(-1, 5, (40, 3)).astype(f32), rng.integers(0, 4, (40, 3)).astype(f32),
                             (rng.integers(0, 8, (40, 3)) * 0.5).astype(f32)])
        for max_range, mds in [(10.0, 0.0), (1.0, 0.0), (1.0, 0.25), (2.0, 1.0), (0.5, 0.0), (10.0, 2.0), (1.5, 2.25)]:
            bad = check(pts, qs, max_range, mds)
            total += bad
            if bad:
                print("trial", trial, "n", n, "kind", kind, "max_range", max_range, "mds", mds, "bad", bad)
    print("mismatches:", total)
