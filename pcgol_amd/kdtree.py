"""pc/storage/kdtree mirror: KDTree on the GPU behind the storage.Search shape
(pc/storage/search.go:13-17): Vec3At / Len / Nearest, plus batched NearestBatch."""
import ctypes as C

import numpy as np

from . import _lib as L
from .pc import PointCloud


class Neighbor:  # pc/storage/search.go:8-11
    __slots__ = ("ID", "DistSq")

    def __init__(self, ID, DistSq):
        self.ID = int(ID)
        self.DistSq = np.float32(DistSq)

    def __repr__(self):
        return "Neighbor{ID:%d, DistSq:%r}" % (self.ID, float(self.DistSq))


class KDTree:
    """kdtree.New(ra, opts...) (kdtree.go:33-56).  `ra`: PointCloud or (n,3) float32 array.
    MinDistSq > 0 makes Nearest the reference's approximate search (kdtree.go:20-22)."""

    def __init__(self, ra, MinDistSq=0.0, _share=None):
        self.MinDistSq = float(MinDistSq)
        if _share is not None:
            self._h, self._owner = _share._h, _share
            return
        if isinstance(ra, PointCloud):
            data, n, s, o = ra.Data, ra.Points, ra.Stride(), ra.xyz_offset()
        else:
            data = L.f32c(ra).reshape(-1, 3)
            n, s, o = len(data), 12, 0
        h = C.c_void_p()
        L.check(L.lib().pcgx_kdtree_build(L.ptr(data), n, s, o, C.byref(h)))
        self._h, self._owner = h, None

    New = classmethod(lambda cls, ra, **opts: cls(ra, **opts))

    def With(self, MinDistSq):  # kdtree.go:59-65: shallow copy with options
        return KDTree(None, MinDistSq=MinDistSq, _share=self if self._owner is None else self._owner)

    def __del__(self):
        if getattr(self, "_owner", 1) is None and getattr(self, "_h", None):
            try:
                L.lib().pcgx_kdtree_free(self._h)
            except Exception:
                pass
            self._h = None

    # -- pc.Vec3RandomAccessor
    def Len(self):
        n = C.c_int64()
        L.check(L.lib().pcgx_kdtree_len(self._h, C.byref(n)))
        return n.value

    def Vec3At(self, i):
        ids = np.array([i], np.int64)
        out = np.empty(3, np.float32)
        L.check(L.lib().pcgx_kdtree_points(self._h, L.ptr(ids), 1, L.ptr(out)))
        return out

    def RawIndexAt(self, i):
        return i

    def MaxDepth(self):
        d = C.c_int32()
        L.check(L.lib().pcgx_kdtree_max_depth(self._h, C.byref(d)))
        return d.value

    def Tree(self):
        """The tree as the reference holds it, nested [id, dim, child0, child1] (None = nil); after
        DeletePoint the patched tree (kdtree.go:264-320)."""
        n = C.c_int64()
        L.check(L.lib().pcgx_kdtree_dump(self._h, None, 0, C.byref(n)))
        d = np.empty((max(n.value, 1), 4), np.int64)
        L.check(L.lib().pcgx_kdtree_dump(self._h, L.ptr(d), n.value, C.byref(n)))

        def rec(k):
            return None if k < 0 else [int(d[k][0]), int(d[k][1]), rec(int(d[k][2])), rec(int(d[k][3]))]
        return rec(0) if n.value else None

    def InOrder(self):
        out = np.empty(self.LiveCount(), np.int64)
        L.check(L.lib().pcgx_kdtree_inorder(self._h, L.ptr(out)))
        return out

    # -- KDTree.DeletePoint (kdtree.go:322-332)
    def DeletePoint(self, pID):
        """Removes point pID from the tree (the accessor, Len() and Vec3At() keep it).  An id
        outside [0, Len()) raises IndexError with the reference's message (kdtree.go:323-325)."""
        self.DeletePoints([pID])

    def DeletePoints(self, ids):
        ids = np.ascontiguousarray(ids, dtype=np.int64).reshape(-1)
        rc = L.lib().pcgx_kdtree_delete_points(self._h, L.ptr(ids), len(ids))
        if rc == L.PCGX_E_OUT_OF_RANGE:
            raise IndexError(L.last_error())
        L.check(rc)

    def LiveCount(self):
        n = C.c_int64()
        L.check(L.lib().pcgx_kdtree_live_count(self._h, C.byref(n)))
        return n.value

    # -- storage.Search
    def Nearest(self, p, maxRange):
        ids, dsq = self.NearestBatch(np.asarray(p, np.float32).reshape(1, 3), maxRange)
        return Neighbor(ids[0], dsq[0])

    def NearestBatch(self, q, maxRange):
        """k.Nearest(q[i], maxRange) for every row of q -> (ids int64[n], distSq float32[n])."""
        q = L.f32c(q).reshape(-1, 3)
        ids = np.empty(len(q), np.int64)
        dsq = np.empty(len(q), np.float32)
        L.check(L.lib().pcgx_kdtree_nearest_batch(self._h, L.ptr(q), len(q), maxRange, self.MinDistSq,
                                                  L.ptr(ids), L.ptr(dsq)))
        return ids, dsq

    def Range(self, p, maxRange):
        """KDTree.Range (kdtree.go:148-161): [Neighbor] with DistSq < maxRange^2 sorted by DistSq."""
        offs, ids, dsq = self.RangeBatch(np.asarray(p, np.float32).reshape(1, 3), maxRange)
        return [Neighbor(i, d) for i, d in zip(ids, dsq)]

    def RangeBatch(self, q, maxRange):
        """Range for every row of q -> (offsets int64[n+1], ids int64[total], distSq float32[total]);
        query i owns [offsets[i], offsets[i+1])."""
        q = L.f32c(q).reshape(-1, 3)
        n = len(q)
        counts = np.zeros(n, np.int64)
        L.check(L.lib().pcgx_kdtree_range_count(self._h, L.ptr(q), n, maxRange, L.ptr(counts)))
        offs = np.zeros(n + 1, np.int64)
        np.cumsum(counts, out=offs[1:])
        total = int(offs[-1])
        ids = np.empty(total, np.int64)
        dsq = np.empty(total, np.float32)
        L.check(L.lib().pcgx_kdtree_range_fill(self._h, L.ptr(q), n, maxRange, L.ptr(offs), L.ptr(ids), L.ptr(dsq)))
        return offs, ids, dsq

    def NearestBatchDev(self, d_q, nq, maxRange, d_ids, d_dsq, presort=True, stream=0):
        """Device-resident variant: raw device addresses (e.g. torch .data_ptr())."""
        L.check(L.lib().pcgx_kdtree_nearest_batch_dev(
            self._h, L.ptr(d_q), nq, maxRange, self.MinDistSq, L.PCGX_KNN_PRESORT if presort else 0,
            L.ptr(d_ids), L.ptr(d_dsq), L.ptr(stream) if stream else None))


New = KDTree.New
