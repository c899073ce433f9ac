"""pc/filter/voxelgrid mirror: voxelgrid.New(leaf, WithChunkSize(..)).Filter(pp)
(voxelgrid.go:23-134) behind filter.Filter (pc/filter/filter.go:7-9)."""
import ctypes as C

import numpy as np

from . import _lib as L
from .pc import PointCloud


def WithChunkSize(s):  # option.go:14-18
    def opt(o):
        o.ChunkSize = [int(v) for v in s]
    return opt


class VoxelGrid:
    def __init__(self, leafSize, *opts):
        self.LeafSize = np.asarray(leafSize, np.float32).reshape(3)
        self.ChunkSize = [0, 0, 0]
        self._scratch = np.empty(0, np.uint8)  # worst-case output buffer, kept like the reference's f.voxels
        for o in opts:
            o(self)

    def Filter(self, pp):
        """Returns a new PointCloud (header cloned, Width = M, Height = 1)."""
        if not isinstance(pp, PointCloud):
            pp = PointCloud.from_xyz(pp)
        stride, off = pp.Stride(), pp.xyz_offset()
        n = pp.Points
        if len(self._scratch) < max(n, 1) * stride:
            self._scratch = np.empty(max(n, 1) * stride, np.uint8)
        out = self._scratch
        m = C.c_int64()
        chunk = np.asarray(self.ChunkSize, np.int32)
        L.check(L.lib().pcgx_voxel_filter(L.ptr(pp.Data), n, stride, off, L.ptr(self.LeafSize),
                                          L.ptr(chunk), L.ptr(out), C.byref(m)))
        h = pp.PointCloudHeader.Clone()
        h.Width, h.Height = m.value, 1
        return PointCloud(h, m.value, out[: m.value * stride].copy())

    def FilterShard(self, pp, comm):
        """This rank's share of Filter(pp) over the ranks of `comm` (collective; every rank passes the
        same cloud): a PointCloud with the records of its contiguous part of the output."""
        if not isinstance(pp, PointCloud):
            pp = PointCloud.from_xyz(pp)
        stride, off = pp.Stride(), pp.xyz_offset()
        n = pp.Points
        out = np.empty(max(n, 1) * stride, np.uint8)
        m = C.c_int64()
        chunk = np.asarray(self.ChunkSize, np.int32)
        L.check(L.lib().pcgx_voxel_filter_sharded(comm._h, L.ptr(pp.Data), n, stride, off, L.ptr(self.LeafSize),
                                                  L.ptr(chunk), L.ptr(out), C.byref(m)))
        h = pp.PointCloudHeader.Clone()
        h.Width, h.Height = m.value, 1
        return PointCloud(h, m.value, out[: m.value * stride].copy())

    def FilterDev(self, d_data, n, stride, off, d_out, stream=0):
        """Device-resident variant (raw device addresses). Returns M.  (The call is half a millisecond at ten million
        points: the argument marshalling is done once per filter object, not per call.)"""
        key = (tuple(self.ChunkSize), self.LeafSize.tobytes())
        if getattr(self, "_dev_args", None) is None or self._dev_args[0] != key:
            leaf = (C.c_float * 3)(*[float(v) for v in self.LeafSize])
            chunk = (C.c_int32 * 3)(*[int(v) for v in self.ChunkSize])
            self._dev_args = (key, leaf, chunk, L.lib().pcgx_voxel_filter_dev)
        _, leaf, chunk, fn = self._dev_args
        m = C.c_int64()  # per call: ctypes releases the GIL, two threads on one filter object must not share the out-parameter
        rc = fn(C.c_void_p(d_data), n, stride, off, leaf, chunk, C.c_void_p(d_out), C.byref(m), C.c_void_p(stream) if stream else None)
        if rc:
            L.check(rc)
        return m.value


    def FilterShardDev(self, comm, d_data, n, stride, off, d_out, stream=0):
        """This rank's share of the filter over the ranks of `comm` (pcgol_amd.distributed.Comm): every
        rank passes the same device-resident cloud and gets the records of its contiguous part of the
        reference's output order; rank 0's, rank 1's, ... one after the other are Filter's output.
        Collective.  Returns this rank's M."""
        m = C.c_int64()
        chunk = np.asarray(self.ChunkSize, np.int32)
        L.check(L.lib().pcgx_voxel_filter_sharded_dev(comm._h, L.ptr(d_data), n, stride, off, L.ptr(self.LeafSize),
                                                      L.ptr(chunk), L.ptr(d_out), C.byref(m),
                                                      L.ptr(stream) if stream else None))
        return m.value


def New(leafSize, *opts):
    return VoxelGrid(leafSize, *opts)
