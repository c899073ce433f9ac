"""CPU check of the certificate behind the grid fast path (csrc/knn_grid.h, grid_cover): every
point whose float32 DistSq to a query is <= L lies, per axis, in the cells
[cell(q - rad), cell(q + rad)] with rad = fl(sqrt(L)) * 1.0001f -- restated here in numpy float32
with the kernel's operation order, on random and adversarial inputs (points a few ulps around the
radius, huge offsets, tiny and huge cells).  No tolerance is involved: the claim rests on float
rounding being monotonic."""
import numpy as np

f32 = np.float32


def cell(v, lo, inv_h, n):
    f = (v.astype(f32) - f32(lo)).astype(f32) * f32(inv_h)
    f = np.fmin(np.fmax(f.astype(f32), f32(0.0)), f32(n - 1))  # fmaxf / fminf: a NaN operand is ignored
    return f.astype(np.int64)  # truncation, as (int)f


def dist_sq(p, q):
    d = (p - q).astype(f32)
    d2 = (d * d).astype(f32)
    return ((d2[:, 0] + d2[:, 1]).astype(f32) + d2[:, 2]).astype(f32)


def check(p, q, lo, h, n, lim=None):
    inv_h = f32(1.0) / f32(h)
    d = dist_sq(p, q)
    L = d if lim is None else lim
    ok = d <= L
    rad = (np.sqrt(L.astype(f32)).astype(f32) * f32(1.0001)).astype(f32)
    for ax in range(3):
        c0 = cell((q[:, ax] - rad).astype(f32), lo[ax], inv_h, n)
        c1 = cell((q[:, ax] + rad).astype(f32), lo[ax], inv_h, n)
        cp = cell(p[:, ax], lo[ax], inv_h, n)
        bad = ok & ((cp < c0) | (cp > c1))
        assert not bad.any(), (ax, p[bad][:3], q[bad][:3], L[bad][:3])


def test_cover_holds_for_random_pairs():
    rng = np.random.default_rng(0)
    for width, h, off in ((10.0, 0.126, 0.0), (10.0, 0.126, 1.0e4), (1.0e-3, 2.0e-5, 0.0), (3.0e5, 700.0, -2.0e6),
                          (40.0, 0.126, 123.456)):
        n = int(width / h) + 1
        lo = np.array([off, off, off], f32)
        q = (rng.uniform(0, width, (400000, 3)) + off).astype(f32)
        # near points (the interesting radius range: up to a few cells)
        p = (q + rng.normal(0, h, (400000, 3))).astype(f32)
        check(p, q, lo, h, n)
        # queries outside the box, points inside
        qo = (rng.uniform(-width, 2 * width, (100000, 3)) + off).astype(f32)
        po = (rng.uniform(0, width, (100000, 3)) + off).astype(f32)
        check(po, qo, lo, h, n)


def test_cover_holds_at_the_radius_to_the_ulp():
    """Points placed exactly on, and a few ulps either side of, q +- sqrt(L) along an axis."""
    rng = np.random.default_rng(1)
    h, n = f32(0.126), 80
    lo = np.zeros(3, f32)
    q = rng.uniform(0, 10, (200000, 3)).astype(f32)
    r = rng.uniform(0.001, 0.3, 200000).astype(f32)
    for ax in range(3):
        for sign in (-1.0, 1.0):
            for ulps in (-3, -1, 0, 1, 3):
                p = q.copy()
                v = (q[:, ax] + f32(sign) * r).astype(f32)
                for _ in range(abs(ulps)):
                    v = np.nextafter(v, f32(np.inf if ulps > 0 else -np.inf)).astype(f32)
                p[:, ax] = v
                check(p, q, lo, h, n)                                   # L = the pair's own DistSq
                check(p, q, lo, h, n, lim=(r * r).astype(f32))          # L = r^2: only pairs within it count


def test_cell_is_monotonic():
    rng = np.random.default_rng(2)
    v = np.sort(rng.uniform(-5, 15, 1_000_000).astype(f32))
    c = cell(v, 0.0, f32(1.0) / f32(0.126), 80)
    assert (np.diff(c) >= 0).all() and c.min() == 0 and c.max() == 79
    assert cell(np.array([np.nan], f32), 0.0, 7.9, 80)[0] in (0,)  # NaN -> cell 0 (fmaxf(NaN, 0) = 0)
