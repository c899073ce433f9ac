"""ctypes front-end of oracle/liboracle.so (the CPU restatement of the reference).

TEST INFRASTRUCTURE ONLY.  Parity pinning: tests/golden/ref_*.json (the
reference's own known-answer tables) via tests/test_oracle_golden.py.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")

ORC_OK, ORC_E_NO_POINT, ORC_E_NOT_ENOUGH_PAIRS, ORC_E_PANIC, ORC_E_OOM, ORC_E_SINGULAR = range(6)


class OracleError(RuntimeError):
    def __init__(self, code):
        self.code = code
        super().__init__({1: "no point", 2: "not enough correspondence pairs",
                          3: "reference would panic (index out of range)",
                          4: "out of memory", 5: "normal equations not positive definite"}.get(code, "oracle error %d" % code))


def build(force=False):
    """Compile liboracle.so with gcc (oracle/Makefile)."""
    srcs = [os.path.join(_HERE, f) for f in ("pcgol_oracle.c", "plane_oracle.c", "segment_oracle.c", "Makefile")]
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"],
                              stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        vp, i64, i32, f32 = C.c_void_p, C.c_int64, C.c_int32, C.c_float
        L.orc_kdtree_new.restype = vp
        L.orc_kdtree_new.argtypes = [vp, i64, i32, i32]
        L.orc_kdtree_free.argtypes = [vp]
        L.orc_kdtree_max_depth.restype = i32
        L.orc_kdtree_max_depth.argtypes = [vp]
        L.orc_kdtree_dump.restype = i64
        L.orc_kdtree_dump.argtypes = [vp, vp]
        L.orc_kdtree_inorder.argtypes = [vp, vp]
        L.orc_kdtree_nearest.argtypes = [vp, vp, f32, f32, vp, vp]
        L.orc_kdtree_nearest_batch.argtypes = [vp, vp, i64, f32, f32, vp, vp, vp, vp]
        L.orc_kdtree_search_leaf.restype = i64
        L.orc_kdtree_search_leaf.argtypes = [vp, vp]
        L.orc_naive_nearest.argtypes = [vp, i64, vp, f32, vp, vp]
        L.orc_kdtree_range.restype = i64
        L.orc_kdtree_range.argtypes = [vp, vp, f32, vp, vp, i64]
        L.orc_kdtree_find_minimum.restype = i64
        L.orc_kdtree_find_minimum.argtypes = [vp, i32]
        L.orc_kdtree_delete_point.restype = i32
        L.orc_kdtree_delete_point.argtypes = [vp, i64]
        L.orc_minmax.restype = i32
        L.orc_minmax.argtypes = [vp, i64, i32, i32, vp, vp]
        L.orc_voxel_filter.restype = i32
        L.orc_voxel_filter.argtypes = [vp, i64, i32, i32, vp, vp, vp, vp]
        L.orc_set_weight_fn.argtypes = [i32, f32]
        L.orc_icp_pairs.restype = i64
        L.orc_icp_pairs.argtypes = [vp, vp, i64, f32, f32, vp, vp, vp]
        L.orc_icp_evaluate.restype = i32
        L.orc_icp_evaluate.argtypes = [vp, vp, i64, f32, f32, i32, i32, vp, vp, vp, vp, vp]
        L.orc_rodrigues.argtypes = [vp, vp]
        L.orc_icp_update.restype = i32
        L.orc_icp_update.argtypes = [vp, vp, i32, vp, vp, vp]
        L.orc_icp_fit.restype = i32
        L.orc_icp_fit.argtypes = [vp, vp, i64, f32, f32, i32, vp, vp, i32, i32, vp, vp, vp, vp, vp]
        L.orc_plane_sums.restype = i32
        L.orc_plane_sums.argtypes = [vp, vp, vp, i64, f32, vp]
        L.orc_plane_finish.restype = i32
        L.orc_plane_finish.argtypes = [vp, i32, vp, vp, vp, vp]
        L.orc_gauss_newton_update.restype = i32
        L.orc_gauss_newton_update.argtypes = [vp, f32, i32, vp, vp, vp, vp]
        L.orc_plane_fit.restype = i32
        L.orc_plane_fit.argtypes = [vp, vp, vp, i64, f32, i32, vp, f32, i32, vp, vp, vp, vp, vp]
        L.orc_grid_new.restype = vp
        L.orc_grid_new.argtypes = [f32, vp, vp]
        L.orc_grid_free.argtypes = [vp]
        L.orc_grid_addr.restype = i32
        L.orc_grid_addr.argtypes = [vp, vp, vp]
        L.orc_grid_add.restype = i32
        L.orc_grid_add.argtypes = [vp, vp, i64]
        L.orc_grid_add_by_addr.argtypes = [vp, i64, i64]
        L.orc_grid_get.restype = i64
        L.orc_grid_get.argtypes = [vp, vp, vp, i64]
        L.orc_grid_get_by_addr.restype = i64
        L.orc_grid_get_by_addr.argtypes = [vp, i64, vp, i64]
        L.orc_grid_indice.restype = i64
        L.orc_grid_indice.argtypes = [vp, vp]
        L.orc_grid_segment.restype = i64
        L.orc_grid_segment.argtypes = [vp, vp, vp]
        L.orc_region_growing_segment.restype = i64
        L.orc_region_growing_segment.argtypes = [vp, vp, vp, f32, vp]
        L.orc_mat4_mul.argtypes = [vp, vp, vp]
        L.orc_mat4_transform.argtypes = [vp, vp, vp]
        L.orc_translate.argtypes = [f32, f32, f32, vp]
        L.orc_rotate.argtypes = [f32, f32, f32, f32, vp]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


# ------------------------------------------------------------------ mat

def translate(x, y, z):
    out = np.empty(16, np.float32)
    lib().orc_translate(x, y, z, _p(out))
    return out


def rotate(x, y, z, ang):
    out = np.empty(16, np.float32)
    lib().orc_rotate(x, y, z, ang, _p(out))
    return out


def mat4_mul(m, a):
    m, a = _f32(m), _f32(a)
    out = np.empty(16, np.float32)
    lib().orc_mat4_mul(_p(m), _p(a), _p(out))
    return out


def mat4_transform(m, pts):
    m = _f32(m)
    pts = _f32(pts).reshape(-1, 3)
    out = np.empty_like(pts)
    L = lib()
    for i in range(len(pts)):
        L.orc_mat4_transform(_p(m), C.c_void_p(pts.ctypes.data + 12 * i),
                             C.c_void_p(out.ctypes.data + 12 * i))
    return out


def rodrigues(v):
    v = _f32(v)
    out = np.empty(16, np.float32)
    lib().orc_rodrigues(_p(v), _p(out))
    return out


# ---------------------------------------------------------------- cloud

def minmax(data, n, stride=12, off=0):
    data = np.ascontiguousarray(data)
    mn, mx = np.empty(3, np.float32), np.empty(3, np.float32)
    rc = lib().orc_minmax(_p(data), n, stride, off, _p(mn), _p(mx))
    if rc:
        raise OracleError(rc)
    return mn, mx


def voxel_filter(data, n, stride, off, leaf, chunk=(0, 0, 0)):
    """Returns the output records as a uint8 array [M*stride]."""
    data = np.ascontiguousarray(data).view(np.uint8).reshape(-1)
    assert data.size >= n * stride
    leaf = _f32(leaf)
    chunk = np.ascontiguousarray(chunk, dtype=np.int32)
    out = np.empty(max(n, 1) * stride, np.uint8)
    m = C.c_int64(0)
    rc = lib().orc_voxel_filter(_p(data), n, stride, off, _p(leaf), _p(chunk), _p(out), C.byref(m))
    if rc:
        raise OracleError(rc)
    return out[: m.value * stride].copy()


# --------------------------------------------------------------- kdtree

class KDTree:
    """pc/storage/kdtree/kdtree.go KDTree (New, Nearest, Range, MinDistSq)."""

    def __init__(self, pts, min_dist_sq=0.0):
        self.pts = _f32(pts).reshape(-1, 3)
        self.n = len(self.pts)
        self.min_dist_sq = float(min_dist_sq)
        self.h = lib().orc_kdtree_new(_p(self.pts), self.n, 12, 0)
        if not self.h:
            raise OracleError(ORC_E_NO_POINT)

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_kdtree_free(self.h)
            self.h = None

    def max_depth(self):
        return lib().orc_kdtree_max_depth(self.h)

    def dump(self):
        out = np.empty((self.n, 4), np.int64)
        k = lib().orc_kdtree_dump(self.h, _p(out))
        return out[:k]

    def inorder(self):
        out = np.empty(self.n, np.int64)
        lib().orc_kdtree_inorder(self.h, _p(out))
        return out

    def nearest(self, p, max_range):
        p = _f32(p)
        i, d = C.c_int64(0), C.c_float(0)
        lib().orc_kdtree_nearest(self.h, _p(p), max_range, self.min_dist_sq, C.byref(i), C.byref(d))
        return i.value, np.float32(d.value)

    def nearest_batch(self, q, max_range, stats=False):
        q = _f32(q).reshape(-1, 3)
        ids = np.empty(len(q), np.int64)
        dsq = np.empty(len(q), np.float32)
        tv, td = C.c_int64(0), C.c_int64(0)
        lib().orc_kdtree_nearest_batch(self.h, _p(q), len(q), max_range, self.min_dist_sq,
                                       _p(ids), _p(dsq), C.byref(tv), C.byref(td))
        if stats:
            return ids, dsq, tv.value, td.value
        return ids, dsq

    def find_minimum(self, dim):
        """findMinimumImpl from the root (kdtree.go:224-262); ValueError for dim > 2."""
        r = lib().orc_kdtree_find_minimum(self.h, dim)
        if r == -2:
            raise ValueError("dim should be <3")
        return r

    def delete_point(self, pid):
        """KDTree.DeletePoint (kdtree.go:322-332); IndexError for an id outside [0, Len())."""
        if lib().orc_kdtree_delete_point(self.h, pid):
            raise IndexError("%d does not correspond to any point in the tree" % pid)

    def tree(self):
        """Nested [id, dim, child0, child1] form of the current tree (None = nil)."""
        d = self.dump()
        def rec(k):
            if k < 0:
                return None
            return [int(d[k][0]), int(d[k][1]), rec(int(d[k][2])), rec(int(d[k][3]))]
        return rec(0) if len(d) else None

    def search_leaf(self, p):
        p = _f32(p)
        return lib().orc_kdtree_search_leaf(self.h, _p(p))

    def range(self, p, max_range, cap=1 << 16):
        p = _f32(p)
        ids = np.empty(cap, np.int64)
        dsq = np.empty(cap, np.float32)
        n = lib().orc_kdtree_range(self.h, _p(p), max_range, _p(ids), _p(dsq), cap)
        assert n <= cap
        return ids[:n].copy(), dsq[:n].copy()


def naive_nearest(pts, p, max_range):
    pts = _f32(pts).reshape(-1, 3)
    p = _f32(p)
    i, d = C.c_int64(0), C.c_float(0)
    lib().orc_naive_nearest(_p(pts), len(pts), _p(p), max_range, C.byref(i), C.byref(d))
    return i.value, np.float32(d.value)


# ------------------------------------------------------------------ icp

def set_weight_fn(kind=0, a=0.0):
    """PointToPointEvaluator.WeightFn for the evaluate / fit calls that follow (0: the default, w = 1;
    1 constant a; 2 1/(a+d); 3 Huber k^2 = a; 4 Tukey c^2 = a -- include/pcgx.h PCGX_WEIGHT_*)."""
    lib().orc_set_weight_fn(int(kind), float(a))


def icp_pairs(tree, target, max_dist):
    target = _f32(target).reshape(-1, 3)
    n = len(target)
    b = np.empty(n, np.int64)
    t = np.empty(n, np.int64)
    d = np.empty(n, np.float32)
    m = lib().orc_icp_pairs(tree.h, _p(target), n, max_dist, tree.min_dist_sq, _p(b), _p(t), _p(d))
    return b[:m].copy(), t[:m].copy(), d[:m].copy()


def icp_evaluate(tree, target, max_dist, min_pairs=0, sums_mode=0):
    """Returns dict(value, gradient[6], dist_rms, npairs, raw10)."""
    target = _f32(target).reshape(-1, 3)
    v, r = C.c_float(0), C.c_float(0)
    g = np.empty(6, np.float32)
    npairs = C.c_int64(0)
    raw = np.zeros(10, np.float64)
    rc = lib().orc_icp_evaluate(tree.h, _p(target), len(target), max_dist, tree.min_dist_sq,
                                min_pairs, sums_mode, C.byref(v), _p(g), C.byref(r),
                                C.byref(npairs), _p(raw))
    if rc:
        raise OracleError(rc)
    return dict(value=np.float32(v.value), gradient=g, dist_rms=np.float32(r.value),
                npairs=npairs.value, raw10=raw)


def icp_update(trans, grad6, it, weight=None, threshold=None, max_iter=0):
    """Returns (trans', converged, it')."""
    w = _f32(weight if weight is not None else np.zeros(6))
    th = _f32(threshold if threshold is not None else np.zeros(6))
    tr = _f32(trans).copy()
    g = _f32(grad6)
    i = C.c_int32(it)
    conv = lib().orc_icp_update(_p(w), _p(th), max_iter, C.byref(i), _p(g), _p(tr))
    return tr, bool(conv), i.value


def icp_fit(tree, target, max_dist, min_pairs=0, weight=None, threshold=None, max_iter=0,
            sums_mode=0):
    """Returns dict(trans[16], value, gradient, dist_rms, num_iteration)."""
    target = _f32(target).reshape(-1, 3)
    w = _f32(weight if weight is not None else np.zeros(6))
    th = _f32(threshold if threshold is not None else np.zeros(6))
    tr = np.empty(16, np.float32)
    v, r = C.c_float(0), C.c_float(0)
    g = np.zeros(6, np.float32)
    nit = C.c_int32(0)
    rc = lib().orc_icp_fit(tree.h, _p(target), len(target), max_dist, tree.min_dist_sq, min_pairs,
                           _p(w), _p(th), max_iter, sums_mode, _p(tr), C.byref(v), _p(g),
                           C.byref(r), C.byref(nit))
    if rc:
        e = OracleError(rc)
        e.trans = tr
        e.num_iteration = nit.value
        raise e
    return dict(trans=tr, value=np.float32(v.value), gradient=g, dist_rms=np.float32(r.value),
                num_iteration=nit.value)


# ------------------------------------------- point-to-plane extension (parity unpinned)

def plane_sums(tree, normals, target, max_dist):
    """The 30 float64 sums of one point-to-plane evaluation (oracle/plane_oracle.c)."""
    normals = _f32(normals).reshape(-1, 3)
    target = _f32(target).reshape(-1, 3)
    assert len(normals) == tree.n
    out = np.zeros(30, np.float64)
    lib().orc_plane_sums(tree.h, _p(normals), _p(target), len(target), max_dist, _p(out))
    return out


def plane_finish(sums30, min_pairs=0):
    sums30 = np.ascontiguousarray(sums30, np.float64)
    v = C.c_float(0)
    g = np.empty(6, np.float32)
    h = np.empty(36, np.float32)
    npairs = C.c_int64(0)
    rc = lib().orc_plane_finish(_p(sums30), min_pairs, C.byref(v), _p(g), _p(h), C.byref(npairs))
    if rc:
        raise OracleError(rc)
    return dict(value=np.float32(v.value), gradient=g, hessian=h, npairs=npairs.value)


def gauss_newton_update(trans, grad6, hess36, it, threshold=None, damping=0.0, max_iter=0):
    """Returns (trans', converged, it'); raises OracleError(ORC_E_SINGULAR)."""
    th = _f32(threshold if threshold is not None else np.zeros(6))
    tr = _f32(trans).copy()
    g, h = _f32(grad6), _f32(hess36)
    i = C.c_int32(it)
    rc = lib().orc_gauss_newton_update(_p(th), damping, max_iter, C.byref(i), _p(g), _p(h), _p(tr))
    if rc < 0:
        raise OracleError(ORC_E_SINGULAR)
    return tr, bool(rc), i.value


def plane_fit(tree, normals, target, max_dist, min_pairs=0, threshold=None, damping=0.0, max_iter=0):
    normals = _f32(normals).reshape(-1, 3)
    target = _f32(target).reshape(-1, 3)
    th = _f32(threshold if threshold is not None else np.zeros(6))
    tr = np.empty(16, np.float32)
    v = C.c_float(0)
    g = np.zeros(6, np.float32)
    h = np.zeros(36, np.float32)
    nit = C.c_int32(0)
    rc = lib().orc_plane_fit(tree.h, _p(normals), _p(target), len(target), max_dist, min_pairs, _p(th), damping,
                             max_iter, _p(tr), C.byref(v), _p(g), _p(h), C.byref(nit))
    if rc:
        e = OracleError(rc)
        e.trans = tr
        e.num_iteration = nit.value
        raise e
    return dict(trans=tr, value=np.float32(v.value), gradient=g, hessian=h, num_iteration=nit.value)


# ------------------------------------ bucket voxel grid / flood fill / region growing

class BucketGrid:
    """pc/storage/voxelgrid.VoxelGrid + pc/segmentation/voxelgrid.Segment (oracle/segment_oracle.c)."""

    def __init__(self, resolution, size, origin):
        self.size = np.ascontiguousarray(size, dtype=np.int64)
        self.origin = _f32(origin)
        self.h = lib().orc_grid_new(resolution, _p(self.size), _p(self.origin))
        self.n = 0

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_grid_free(self.h)
            self.h = None

    def add(self, p, index):
        p = _f32(p)
        ok = bool(lib().orc_grid_add(self.h, _p(p), index))
        self.n += ok
        return ok

    def add_all(self, pts):
        for i, p in enumerate(_f32(pts).reshape(-1, 3)):
            self.add(p, i)

    def addr(self, p):
        p = _f32(p)
        a = C.c_int64(0)
        ok = lib().orc_grid_addr(self.h, _p(p), C.byref(a))
        return (a.value, True) if ok else (0, False)

    def add_by_addr(self, a, index):
        lib().orc_grid_add_by_addr(self.h, a, index)
        self.n += 1

    def get(self, p):
        p = _f32(p)
        out = np.empty(max(self.n, 1), np.int64)
        k = lib().orc_grid_get(self.h, _p(p), _p(out), len(out))
        return None if k < 0 else out[:k].copy()

    def indice(self):
        out = np.empty(max(self.n, 1), np.int64)
        k = lib().orc_grid_indice(self.h, _p(out))
        return out[:k].copy()

    def segment(self, p):
        p = _f32(p)
        out = np.empty(max(self.n, 1), np.int64)
        k = lib().orc_grid_segment(self.h, _p(p), _p(out))
        return out[:k].copy()


def region_growing_segment(tree, labels, p, max_range):
    """regiongrowing.Segment (regiongrowing.go:23-56): ids in the reference's BFS order."""
    labels = np.ascontiguousarray(labels, dtype=np.uint32)
    assert len(labels) == tree.n
    p = _f32(p)
    out = np.empty(max(tree.n, 1), np.int64)
    k = lib().orc_region_growing_segment(tree.h, _p(labels), _p(p), max_range, _p(out))
    return out[:k].copy()
