"""mat.Mat4 / Vec3 helpers the hot path uses (mat/mat4.go, mat/transform.go),
evaluated by the library's float32 restatement (host side of libpcgx.so)."""
import numpy as np

from . import _lib as L


def Translate(x, y, z):  # mat/transform.go:7-14
    m = np.eye(4, dtype=np.float32).reshape(-1)
    m[12:15] = (x, y, z)
    return m


def Mul(m, a):  # mat/mat4.go:16-28
    m, a = L.f32c(m), L.f32c(a)
    out = np.empty(16, np.float32)
    L.check(L.lib().pcgx_mat4_mul(L.ptr(m), L.ptr(a), L.ptr(out)))
    return out


def Transform(m, xyz):  # mat/mat4.go:130-137, batched
    m = L.f32c(m)
    xyz = L.f32c(xyz).reshape(-1, 3)
    out = np.empty_like(xyz)
    L.check(L.lib().pcgx_mat4_transform(L.ptr(m), L.ptr(xyz), len(xyz), L.ptr(out)))
    return out


def RodriguesToRotation(v):  # icp/rodrigues.go:11-33
    v = L.f32c(v)
    out = np.empty(16, np.float32)
    L.check(L.lib().pcgx_rodrigues(L.ptr(v), L.ptr(out)))
    return out
