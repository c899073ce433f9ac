"""Thin mirror of the reference's cloud model (pc/pointcloud.go) -- only what
the hot path needs: the AoS record layout contract and xyz field discovery."""
import numpy as np

from . import _lib as L


class PointCloudHeader:
    """pc/pointcloud.go:9-18.  Fields/Size/Type/Count/Width/Height/Viewpoint."""

    def __init__(self, Fields, Size, Count, Type=None, Width=0, Height=1, Version=0.7, Viewpoint=None):
        self.Version = Version
        self.Fields = list(Fields)
        self.Size = list(Size)
        self.Type = list(Type) if Type is not None else ["F"] * len(self.Fields)
        self.Count = list(Count)
        self.Width = Width
        self.Height = Height
        self.Viewpoint = list(Viewpoint) if Viewpoint is not None else []

    def Clone(self):  # pointcloud.go:20-31
        return PointCloudHeader(self.Fields, self.Size, self.Count, self.Type, self.Width, self.Height,
                                self.Version, self.Viewpoint)

    def Stride(self):  # pointcloud.go:64-70
        return sum(c * s for c, s in zip(self.Count, self.Size))


class PointCloud:
    """pc/pointcloud.go:72-78: header + Points + Data ([]byte, AoS little-endian records)."""

    def __init__(self, header, points, data):
        self.PointCloudHeader = header
        self.Points = int(points)
        self.Data = np.ascontiguousarray(data).view(np.uint8).reshape(-1)

    def Stride(self):
        return self.PointCloudHeader.Stride()

    def xyz_offset(self):
        """Byte offset of the xyz triple: one field "xyz" or three consecutive
        fields x, y, z (Vec3Iterator, pointcloud.go:130-150).  Anything else is
        "invalid field name" (pointcloud.go:115,187)."""
        h = self.PointCloudHeader
        off = 0
        state = 0
        start = None
        for name, size, count in zip(h.Fields, h.Size, h.Count):
            if name == "xyz":
                return off
            if name == "x" and state == 0:
                state, start = 1, off
            elif name == "y" and state == 1:
                state = 2
            elif name == "z" and state == 2:
                return start
            else:
                state = 0
            off += size * count
        raise L.ErrInvalidField(L.PCGX_E_BAD_FIELD, "invalid field name")

    def Vec3(self):
        """All points as an (n,3) float32 array (copy)."""
        s, o = self.Stride(), self.xyz_offset()
        rec = self.Data[: self.Points * s].reshape(self.Points, s)
        return np.ascontiguousarray(rec[:, o:o + 12]).view(np.float32).reshape(self.Points, 3)

    @staticmethod
    def from_xyz(xyz):
        xyz = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
        h = PointCloudHeader(["x", "y", "z"], [4, 4, 4], [1, 1, 1], Width=len(xyz), Height=1)
        return PointCloud(h, len(xyz), xyz)


def MinMaxVec3(points):
    """pc.MinMaxVec3 (pc/minmax.go:9-26) on the GPU.  `points`: PointCloud or (n,3) array."""
    if isinstance(points, PointCloud):
        data, n, s, o = points.Data, points.Points, points.Stride(), points.xyz_offset()
    else:
        data = L.f32c(points).reshape(-1, 3)
        n, s, o = len(data), 12, 0
    mn, mx = np.empty(3, np.float32), np.empty(3, np.float32)
    L.check(L.lib().pcgx_minmax(L.ptr(data), n, s, o, L.ptr(mn), L.ptr(mx)))
    return mn, mx
