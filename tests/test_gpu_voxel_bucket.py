"""The VoxelGrid filter's bucket path (csrc/voxel_bucket.hip: the coordinates travel with the sort keys, one workgroup
per bucket of cells) against the oracle, byte for byte -- forced onto clouds far smaller than it is meant for, so that
every branch runs: one and two partition passes, records with fields around xyz (the first point's record is looked
up through its index), unaligned records, chunked keys, empty buckets in front of / behind the occupied ones; and the
clouds it gives up on (a bucket or a cell too crowded for its LDS tile), which the radix path answers."""
import ctypes as C

import numpy as np
import pytest

import oracle as O
from pcgol_amd import _lib as L
from pcgol_amd import synth

pytestmark = pytest.mark.gpu
f32 = np.float32


def _stats(reset=True):
    st = np.zeros(4, np.int64)
    L.check(L.lib().pcgx_debug_voxel_stats(L.ptr(st), 1 if reset else 0))
    return st


def _filter(rec, n, stride, off, leaf, chunk):
    out = np.empty(n * stride, np.uint8)
    m = C.c_int64()
    leafv, chunkv = np.asarray(leaf, f32), np.asarray(chunk, np.int32)
    L.check(L.lib().pcgx_voxel_filter(L.ptr(rec), n, stride, off, L.ptr(leafv), L.ptr(chunkv), L.ptr(out), C.byref(m)))
    return out[: m.value * stride]


def _records(pts, stride, off, rng):
    n = len(pts)
    rec = rng.integers(0, 256, size=(n, stride), dtype=np.uint8)
    rec[:, off:off + 12] = pts.view(np.uint8).reshape(n, 12)
    return np.ascontiguousarray(rec)


@pytest.mark.parametrize("n,width,leaf,chunk,stride,off", [
    (3000, 1.6, 0.05, (0, 0, 0), 12, 0),          # one partition pass, a handful of buckets
    (100000, 1.6, 0.05, (0, 0, 0), 12, 0),        # BASELINE config C1
    (100000, 1.6, 0.05, (0, 0, 0), 16, 0),        # x y z label: the first point's record through its index
    (50000, 1.6, 0.05, (0, 0, 0), 20, 4),
    (50000, 1.6, 0.05, (0, 0, 0), 15, 1),         # unaligned records
    (100000, 1.6, 0.05, (8, 8, 8), 12, 0),        # chunk id and cell in one key
    (100000, 1.6, 0.05, (3, 5, 7), 16, 0),
    (300000, 3.0, 0.02, (0, 0, 0), 12, 0),        # two partition passes
    (300000, 3.0, 0.02, (64, 64, 64), 12, 0),
    (200000, 0.6, 0.02, (0, 0, 0), 12, 0),        # ~7 points per cell: cells beyond the eight-register sort
])
def test_bucket_path_vs_oracle(n, width, leaf, chunk, stride, off, monkeypatch):
    monkeypatch.setenv("PCGX_VOXEL_BUCKET_MIN_N", "1")
    rng = np.random.default_rng(n + stride)
    pts = synth.uniform_cloud(n, width, 40 + n % 97)
    rec = _records(pts, stride, off, rng)
    exp = O.voxel_filter(rec, n, stride, off, (leaf,) * 3, chunk)
    _stats()
    got = _filter(rec, n, stride, off, (leaf,) * 3, chunk)
    st = _stats()
    assert st[0] == 1 and st[1] == 0, st      # the bucket path answered
    assert len(got) == len(exp) and np.array_equal(got, exp)


def test_bucket_path_slab_and_offset_clouds(monkeypatch):
    """Most of the key range empty (a thin slab; a cloud far from the origin of a chunked grid): empty buckets in
    front of, between and behind the occupied ones, the last bucket among them."""
    monkeypatch.setenv("PCGX_VOXEL_BUCKET_MIN_N", "1")
    pts = synth.uniform_cloud(200_000, 3.0, 5)
    slab = pts.copy()
    slab[:, 2] = slab[:, 2] * f32(0.001)
    off_cloud = synth.uniform_cloud(50000, 2.0, 9) - f32(0.7)
    for data, leaf, chunk in ((np.ascontiguousarray(slab), (0.004,) * 3, (0, 0, 0)),
                              (off_cloud, (0.05, 0.04, 0.03), (16, 16, 16)),
                              (synth.uniform_cloud(50000, 1.0, 11) + f32(0.5), (0.05,) * 3, (0, 0, 0))):
        exp = O.voxel_filter(data, len(data), 12, 0, leaf, chunk)
        _stats()
        got = _filter(data, len(data), 12, 0, leaf, chunk)
        st = _stats()
        assert st[0] + st[1] == 1, st     # attempted; whichever path answered, the bytes are the reference's
        assert len(got) == len(exp) and np.array_equal(got, exp), (leaf, chunk, st)


def test_crowded_clouds_fall_back_to_the_radix_path(monkeypatch):
    """A cell with many points (300 copies of one point: the float32 sum over them in input order, inside the
    bucket's LDS tile) is the bucket path's own; a cloud whose points sit in two tight clusters of a large box (buckets
    far over the LDS tile) is what it gives up on -- found on the device, the radix path answers.  The bytes are the
    oracle's either way."""
    monkeypatch.setenv("PCGX_VOXEL_BUCKET_MIN_N", "1")
    pts = synth.uniform_cloud(100_000, 1.6, 3)
    crowded_cell = pts.copy()
    crowded_cell[5000:5300] = crowded_cell[4999]
    crowded_cell[20_000:21_000:2] = crowded_cell[19_999]   # and one whose points alternate with others' in input order
    exp = O.voxel_filter(crowded_cell, len(pts), 12, 0, (0.05,) * 3, (0, 0, 0))
    _stats()
    got = _filter(crowded_cell, len(pts), 12, 0, (0.05,) * 3, (0, 0, 0))
    st = _stats()
    assert st[0] == 1 and st[1] == 0, st
    assert np.array_equal(got, exp)
    rng = np.random.default_rng(7)
    a = (rng.random((150_000, 3)) * 0.05).astype(f32)
    b = (rng.random((150_000, 3)) * 0.05).astype(f32) + f32(7.9)
    clusters = np.ascontiguousarray(np.concatenate([a, b])[rng.permutation(300_000)])
    exp = O.voxel_filter(clusters, len(clusters), 12, 0, (0.02,) * 3, (0, 0, 0))
    _stats()
    got = _filter(clusters, len(clusters), 12, 0, (0.02,) * 3, (0, 0, 0))
    st = _stats()
    assert st[0] == 0 and st[1] == 1 and (st[2] & 1), st
    assert np.array_equal(got, exp)


def test_c3_takes_the_bucket_path():
    """BASELINE config C3 at full size goes the bucket path, plain and chunked (its bytes are compared with the
    oracle's in test_gpu_voxel.py::test_c3_full_size_against_the_oracle)."""
    c = synth.c3_voxel()
    pts = c["points"]
    for chunk in ((0, 0, 0), (64, 64, 64)):
        _stats()
        got = _filter(pts, len(pts), 12, 0, c["leaf"], chunk)
        st = _stats()
        assert st[0] == 1 and st[1] == 0, (chunk, st)
        assert 3_000_000 < len(got) // 12 < 3_400_000


def _last_error():
    buf = C.create_string_buffer(512)
    L.lib().pcgx_last_error(buf, 512)
    return buf.value


def test_the_plan_made_on_the_device_fails_as_the_hosts_does(monkeypatch):
    """The bucket path's grid and plan are made by the min/max launch's last workgroup (csrc/voxel_key.h: one piece of
    code for host and device).  A grid the reference cannot address must end the call with the same status and the same
    words whichever side found it; a point outside the dense grid (negative vMin, non-chunked: the Go code panics) is
    found by the key kernel on either path."""
    pts = synth.uniform_cloud(5000, 1.0, 5)
    neg = pts - f32(10.0)
    cases = [(pts, (1e-7,) * 3, (0, 0, 0)),            # (xs+1)(ys+1)(zs+1) >= 2^32
             (pts, (1e-6,) * 3, (1, 1, 1)),            # chunk grid of >= 2^32 chunks
             (neg, (0.05,) * 3, (0, 0, 0))]            # a key out of range
    for data, leaf, chunk in cases:
        seen = []
        for min_n in ("1", "1000000000"):              # the device's plan / the host's
            monkeypatch.setenv("PCGX_VOXEL_BUCKET_MIN_N", min_n)
            out = np.empty(len(data) * 12, np.uint8)
            m = C.c_int64()
            leafv, chunkv = np.asarray(leaf, f32), np.asarray(chunk, np.int32)
            rc = L.lib().pcgx_voxel_filter(L.ptr(data), len(data), 12, 0, L.ptr(leafv), L.ptr(chunkv), L.ptr(out), C.byref(m))
            seen.append((rc, _last_error()))
        assert seen[0][0] == seen[1][0] == 7 and seen[0][1] == seen[1][1], (leaf, chunk, seen)


def test_calls_the_device_plan_turns_away_go_the_radix_path(monkeypatch):
    """Not a call for the bucket path by what only the device knows when the launches are made -- sort keys of more than
    26 bits (a sparse cloud on a fine grid), the two-sort layout: every kernel behind the plan returns at once, neither
    counter moves, and the radix path answers from the min / max the attempt read back."""
    monkeypatch.setenv("PCGX_VOXEL_BUCKET_MIN_N", "1")
    pts = synth.uniform_cloud(20000, 4.0, 13)
    exp = O.voxel_filter(pts, len(pts), 12, 0, (0.004,) * 3, (0, 0, 0))      # 1000^3 cells: 30 bits
    _stats()
    got = _filter(pts, len(pts), 12, 0, (0.004,) * 3, (0, 0, 0))
    st = _stats()
    assert st[0] == 0 and st[1] == 0, st
    assert np.array_equal(got, exp)
    monkeypatch.setenv("PCGX_VOXEL_TWO_SORTS", "1")
    pts = synth.uniform_cloud(60000, 3.0, 77) - f32(0.7)
    exp = O.voxel_filter(pts, len(pts), 12, 0, (0.05, 0.04, 0.06), (7, 5, 9))
    got = _filter(pts, len(pts), 12, 0, (0.05, 0.04, 0.06), (7, 5, 9))
    st = _stats()
    assert st[0] == 0 and st[1] == 0, st
    assert np.array_equal(got, exp)


@pytest.mark.parametrize("leaf,chunk", [(0.02, (0, 0, 0)), (0.013, (0, 0, 0)), (0.05, (8, 8, 8)), (0.0371, (5, 3, 4)),
                                        ((0.02, 0.031, 0.0173), (0, 0, 0))])
@pytest.mark.parametrize("min_n", ["1", "100000000"])
def test_points_on_and_next_to_cell_faces(leaf, chunk, min_n, monkeypatch):
    """The cell of a point is int(float32 quotient); the kernels take the quotient's integer part from a product with
    the size's reciprocal where that proves it and divide where it does not (voxel_key.h, cell_by_reciprocal).  A cloud
    made to sit ON the faces: coordinates k * leaf (as float64, rounded to float32, and from float32 products) and their
    neighbours up to three steps of the float32 grid to either side -- every one decided as the reference decides it.
    Both paths (the bucket path forced, and switched off)."""
    monkeypatch.setenv("PCGX_VOXEL_BUCKET_MIN_N", min_n)
    rng = np.random.default_rng(23)
    lf = np.asarray(leaf if isinstance(leaf, tuple) else (leaf,) * 3, np.float64)
    n = 120_000
    k = rng.integers(0, 160, size=(n, 3))
    exact = (k * lf).astype(f32)                                       # k * leaf in float64, rounded once
    prod = (k.astype(f32) * lf.astype(f32)).astype(f32)                # the float32 product
    pts = np.where(rng.random((n, 3)) < 0.5, exact, prod).astype(f32)
    steps = rng.integers(-3, 4, size=(n, 3))
    for s in range(1, 4):                                              # up to three steps of the float32 grid away
        pts = np.where(steps >= s, np.nextafter(pts, f32(np.inf)), pts)
        pts = np.where(steps <= -s, np.nextafter(pts, f32(-np.inf)), pts)
    pts = np.ascontiguousarray(np.abs(pts).astype(f32))
    pts[0] = 0.0                                                       # vMin = 0: the faces are where the cloud was put
    leafs = tuple(float(v) for v in lf)
    exp = O.voxel_filter(pts, n, 12, 0, leafs, chunk)
    got = _filter(pts, n, 12, 0, leafs, chunk)
    assert len(got) == len(exp) and np.array_equal(got, exp)
    shifted = np.ascontiguousarray((pts + f32(0.37)).astype(f32))     # vMin != 0: the subtraction's rounding in front
    exp = O.voxel_filter(shifted, n, 12, 0, leafs, chunk)
    got = _filter(shifted, n, 12, 0, leafs, chunk)
    assert len(got) == len(exp) and np.array_equal(got, exp)


@pytest.mark.parametrize("seed", range(24))
def test_random_clouds_and_grids(seed):
    """Clouds and grids drawn at random -- uniform boxes, slabs, clusters of very different density in one cloud (crowded
    cells next to empty buckets; some clouds the bucket path hands to the radix path), leaf sizes per axis, chunks,
    records with fields around xyz -- byte for byte against the oracle, whichever path answers (the library's own
    threshold decides: every cloud here is over it)."""
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(1500, 180_000))
    kind = seed % 4
    if kind == 0:
        pts = (rng.random((n, 3)) * rng.uniform(0.5, 6.0, 3)).astype(f32)
    elif kind == 1:                                               # a slab: one axis a few cells thick
        pts = (rng.random((n, 3)) * np.array([rng.uniform(2, 8), rng.uniform(2, 8), rng.uniform(0.01, 0.2)])).astype(f32)
    elif kind == 2:                                               # clusters of different spread, a sparse background
        k = int(rng.integers(2, 7))
        centers = rng.random((k, 3)) * 5.0
        spread = 10.0 ** rng.uniform(-2.5, -0.3, k)
        which = rng.integers(0, k, n)
        pts = (centers[which] + rng.normal(size=(n, 3)) * spread[which, None]).astype(f32)
        pts[: n // 10] = (rng.random((n // 10, 3)) * 5.0).astype(f32)
    else:                                                         # far from the origin, negative coordinates
        pts = (rng.random((n, 3)) * rng.uniform(0.5, 3.0, 3) + rng.uniform(-500, 500, 3)).astype(f32)
    pts = np.ascontiguousarray(pts[rng.permutation(n)])
    leaf = tuple(float(v) for v in (10.0 ** rng.uniform(-2.0, -0.7, 3) if seed % 3 else np.repeat(10.0 ** rng.uniform(-2.0, -0.7), 3)))
    chunk = (0, 0, 0) if seed % 2 else tuple(int(v) for v in rng.integers(2, 40, 3))
    # (without chunks the reference's dense array is sized by vMax, voxelgrid.go:46: far from the origin that is 10^11
    # voxels, which the oracle would allocate and scan as the Go code would -- such clouds are filtered in chunks)
    if chunk == (0, 0, 0) and abs(float(np.prod(np.floor(pts.max(0) / np.asarray(leaf, f32)) + 1.0))) > 2e7:
        chunk = tuple(int(v) for v in rng.integers(2, 40, 3))
    stride, off = [(12, 0), (16, 0), (20, 4), (15, 1)][int(rng.integers(0, 4))]
    rec = _records(pts, stride, off, rng)
    try:
        exp = O.voxel_filter(rec, n, stride, off, leaf, chunk)
    except O.OracleError:                                          # the Go code would panic (a grid sized by a negative vMax,
        with pytest.raises(L.PcgxError) as ei:                     # voxelgrid.go:46, an index out of range): an error here
            _filter(rec, n, stride, off, leaf, chunk)
        assert ei.value.code == L.PCGX_E_OUT_OF_RANGE, ei.value
        return
    _stats()
    got = _filter(rec, n, stride, off, leaf, chunk)
    st = _stats()
    assert st[0] + st[1] == 1, st
    assert len(got) == len(exp) and np.array_equal(got, exp), (seed, n, leaf, chunk, stride, off, st)
