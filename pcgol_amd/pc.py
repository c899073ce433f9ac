"""Thin mirror of the reference's cloud model (pc/pointcloud.go) -- only what
the hot path needs: the AoS record layout contract and xyz field discovery -- and of the PCD
reader / writer (pc/io.go) over the C ABI."""
import ctypes as C

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


# ------------------------------------------------------------------ PCD (pc/io.go)
Ascii, Binary, BinaryCompressed = 0, 1, 2  # pc.Format (io.go:16-22)


def _header_from_c(h):
    n = h.n_fields
    return PointCloudHeader([bytes(h.fields[i].value).decode("latin-1") for i in range(n)], list(h.size[:n]),
                            list(h.count[:n]), [h.type[i:i + 1].decode("latin-1") for i in range(n)],
                            Width=int(h.width), Height=int(h.height), Version=np.float32(h.version),
                            Viewpoint=[np.float32(v) for v in h.viewpoint[: h.n_viewpoint]])


def _header_to_c(hd, points):
    h = L.PcdHeader()
    h.version = float(hd.Version)
    h.n_fields = len(hd.Fields)
    if h.n_fields > L.PCGX_PCD_MAX_FIELDS:
        raise ValueError("more than %d fields" % L.PCGX_PCD_MAX_FIELDS)
    for i, (f, s, t, c) in enumerate(zip(hd.Fields, hd.Size, hd.Type, hd.Count)):
        h.fields[i].value = f.encode("latin-1")
        h.size[i], h.count[i] = s, c
    h.type = "".join((t or "?")[0] for t in hd.Type).encode("latin-1")
    h.width, h.height, h.points = hd.Width, hd.Height, points
    h.n_viewpoint = len(hd.Viewpoint)
    for i, v in enumerate(hd.Viewpoint):
        h.viewpoint[i] = float(v)
    h.format = Binary
    h.stride = hd.Stride()
    return h


def _parse_header(buf):
    b = np.frombuffer(bytes(buf), np.uint8)
    h = L.PcdHeader()
    L.check(L.lib().pcgx_pcd_unmarshal_header(L.ptr(b) if len(b) else None, len(b), C.byref(h)))
    return b, h


def UnmarshalHeader(buf):
    """pc.UnmarshalHeader (io.go:24-31)."""
    return _header_from_c(_parse_header(buf)[1])


def Unmarshal(buf):
    """pc.Unmarshal (io.go:33-45): ascii / binary / binary_compressed -> PointCloud (host)."""
    b, h = _parse_header(buf)
    data = np.zeros(max(h.points * h.stride, 1), np.uint8)
    L.check(L.lib().pcgx_pcd_unmarshal(L.ptr(b), len(b), C.byref(h), L.ptr(data)))
    return PointCloud(_header_from_c(h), h.points, data[: h.points * h.stride])


def UnmarshalDev(buf, d_out=None, stream=0):
    """Records straight into device memory.  d_out: device address with Points*Stride bytes, or None
    to let torch allocate a uint8 tensor.  Returns (PointCloudHeader, points, stride, tensor or None)."""
    b, h = _parse_header(buf)
    t = None
    if d_out is None:
        import torch
        t = torch.empty(max(h.points * h.stride, 1), dtype=torch.uint8, device="cuda")
        d_out = t.data_ptr()
    L.check(L.lib().pcgx_pcd_unmarshal_dev(L.ptr(b), len(b), C.byref(h), L.ptr(int(d_out)),
                                           L.ptr(stream) if stream else None))
    return _header_from_c(h), int(h.points), int(h.stride), t


def Marshal(pp):
    """pc.Marshal (io.go:232-285): bytes of a "DATA binary" file."""
    h = _header_to_c(pp.PointCloudHeader, pp.Points)
    n = C.c_size_t()
    data = np.ascontiguousarray(pp.Data)
    L.check(L.lib().pcgx_pcd_marshal(C.byref(h), L.ptr(data) if len(data) else None, None, 0, C.byref(n)))
    out = np.empty(n.value, np.uint8)
    L.check(L.lib().pcgx_pcd_marshal(C.byref(h), L.ptr(data) if len(data) else None, L.ptr(out), len(out), C.byref(n)))
    return out.tobytes()
