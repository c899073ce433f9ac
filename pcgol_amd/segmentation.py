"""Mirrors of pc/storage/voxelgrid (bucket grid), pc/segmentation/voxelgrid (flood fill) and
pc/segmentation/regiongrowing on the GPU (include/pcgx.h "bucket voxel grid + segmentation").

The reference fills a VoxelGrid with Add(p, index) calls one by one; the batch seam here is the
whole cloud: StorageVoxelGrid(resolution, size, origin).AddAll(cloud) == Add(point i, i) for all i.
Result order of the Segment calls: see include/pcgx.h (set-equal to the reference's BFS order)."""
import ctypes as C

import numpy as np

from . import _lib as L
from .pc import PointCloud


class StorageVoxelGrid:
    """pc/storage/voxelgrid.VoxelGrid (voxelgrid.go:7-122)."""

    def __init__(self, resolution, size, origin):
        self.resolution = float(resolution)
        self.size = np.ascontiguousarray(size, dtype=np.int64)
        self.origin = L.f32c(origin)
        self._h = None
        self._n = 0
        self.AddAll(np.zeros((0, 3), np.float32))

    def Resolution(self):
        return np.float32(self.resolution)

    def MinMax(self):  # voxelgrid.go:25-31
        ext = self.size.astype(np.float32) * np.float32(self.resolution)
        return self.origin.copy(), self.origin + ext

    def AddAll(self, cloud):
        """Reset() + Add(point i, i) for every point (PointCloud or (n,3) float32)."""
        if isinstance(cloud, PointCloud):
            data, n, s, o = cloud.Data, cloud.Points, cloud.Stride(), cloud.xyz_offset()
        else:
            data = L.f32c(cloud).reshape(-1, 3)
            n, s, o = len(data), 12, 0
        self._free()
        h = C.c_void_p()
        L.check(L.lib().pcgx_bucket_grid_build(L.ptr(data) if n else None, n, s, o, self.resolution, L.ptr(self.size),
                                               L.ptr(self.origin), C.byref(h)))
        self._h, self._n = h, n
        return self.AddedMask()

    def _free(self):
        if getattr(self, "_h", None):
            L.lib().pcgx_bucket_grid_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self._free()
        except Exception:
            pass

    def _counts(self):
        a, b, c = C.c_int64(), C.c_int64(), C.c_int64()
        L.check(L.lib().pcgx_bucket_grid_counts(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def Len(self):  # voxelgrid.go:110-112
        return self._counts()[0]

    def PointAddrs(self):
        out = np.empty(self._n, np.int64)
        L.check(L.lib().pcgx_bucket_grid_point_addrs(self._h, L.ptr(out) if self._n else None))
        return out

    def AddedMask(self):
        """What Add returned for every point (voxelgrid.go:37-45)."""
        return self.PointAddrs() >= 0

    def Addr(self, p):  # voxelgrid.go:64-79
        p = L.f32c(p)
        a, ok = C.c_int64(), C.c_int32()
        L.check(L.lib().pcgx_bucket_grid_addr(self._h, L.ptr(p), C.byref(a), C.byref(ok)))
        return a.value, bool(ok.value)

    def GetByAddr(self, a):
        cnt = C.c_int64()
        L.check(L.lib().pcgx_bucket_grid_get_by_addr(self._h, int(a), None, 0, C.byref(cnt)))
        out = np.empty(max(cnt.value, 1), np.int64)
        L.check(L.lib().pcgx_bucket_grid_get_by_addr(self._h, int(a), L.ptr(out), len(out), C.byref(cnt)))
        return out[: cnt.value]

    def Get(self, p):
        """ids of p's voxel; None (nil) when p is outside the grid (voxelgrid.go:52-58)."""
        a, ok = self.Addr(p)
        return self.GetByAddr(a) if ok else None

    def Indice(self):  # voxelgrid.go:114-120
        out = np.empty(max(self._counts()[1], 1), np.int64)
        L.check(L.lib().pcgx_bucket_grid_indice(self._h, L.ptr(out)))
        return out[: self._counts()[1]]


class SegmentationVoxelGrid(StorageVoxelGrid):
    """pc/segmentation/voxelgrid.VoxelGrid (embeds the storage grid, voxelgrid.go:27-37)."""

    def Storage(self):
        return self

    def Components(self):
        """Segment() for every seed at once: per point the smallest voxel address of its
        26-connected set of occupied voxels (-1 outside the grid)."""
        out = np.empty(self._n, np.int64)
        L.check(L.lib().pcgx_bucket_grid_components(self._h, L.ptr(out) if self._n else None))
        return out

    def Segment(self, p, order="bfs"):
        """segmentation/voxelgrid/voxelgrid.go:39-73.  order="bfs": ids exactly as the reference
        appends them (its FIFO flood fill, run over the device-built buckets); order="address":
        the same set from the device's component labels, ascending voxel address (fast path)."""
        fn = L.lib().pcgx_bucket_grid_segment_bfs if order == "bfs" else L.lib().pcgx_bucket_grid_segment
        p = L.f32c(p)
        cnt = C.c_int64()
        L.check(fn(self._h, L.ptr(p), None, 0, C.byref(cnt)))
        out = np.empty(max(cnt.value, 1), np.int64)
        L.check(fn(self._h, L.ptr(p), L.ptr(out), len(out), C.byref(cnt)))
        return out[: cnt.value]


class RegionGrowing:
    """pc/segmentation/regiongrowing.RegionGrowing (regiongrowing.go:13-56): New(search, propertyIter).
    search: a pcgol_amd KDTree; propertyIter: uint32 value per point (Uint32At)."""

    def __init__(self, search, propertyIter):
        self.search = search
        self.labels = np.ascontiguousarray(propertyIter, dtype=np.uint32)
        if len(self.labels) != search.Len():
            raise ValueError("propertyIter must hold one value per point of search")
        self._comp = {}

    New = classmethod(lambda cls, search, propertyIter: cls(search, propertyIter))

    def Components(self, maxRange):
        """Segment() for every seed at once (cached per maxRange): per point the smallest id of its
        region."""
        key = float(np.float32(maxRange))
        if key not in self._comp:
            out = np.empty(len(self.labels), np.int64)
            L.check(L.lib().pcgx_region_growing_components(self.search._h, L.ptr(self.labels), key, L.ptr(out)))
            self._comp[key] = out
        return self._comp[key]

    def Segment(self, p, maxRange, order="bfs"):
        """regiongrowing.go:23-56.  order="bfs": ids exactly as the reference appends them (its FIFO
        search, one device Range batch per BFS level); order="id": the same set from the device's
        region labels (Components), ascending id (fast path for many seeds)."""
        p = L.f32c(p)
        out = np.empty(max(len(self.labels), 1), np.int64)
        cnt = C.c_int64()
        if order == "bfs":
            L.check(L.lib().pcgx_region_growing_segment_bfs(self.search._h, L.ptr(self.labels), L.ptr(p),
                                                            float(np.float32(maxRange)), L.ptr(out), len(out),
                                                            C.byref(cnt)))
            return out[: cnt.value].copy()
        comp = self.Components(maxRange)
        L.check(L.lib().pcgx_region_growing_segment(self.search._h, L.ptr(self.labels), L.ptr(comp), L.ptr(p),
                                                    float(np.float32(maxRange)), L.ptr(out), len(out), C.byref(cnt)))
        return out[: cnt.value].copy()
