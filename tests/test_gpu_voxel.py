"""GPU parity: VoxelGrid filter and MinMax (through the C ABI) vs the oracle and
the reference's known-answer table.  Output records must be byte-identical."""
import numpy as np
import pytest

import oracle as O
from pcgol_amd import PcgxError, ErrNoPoint, pc, synth, voxelgrid

pytestmark = pytest.mark.gpu
f32 = np.float32


def _golden_cloud(g):
    c = g["cloud"]
    rec = np.zeros((len(c["xyz"]), 4), f32)
    rec[:, :3] = np.array(c["xyz"], f32)
    rec.view(np.uint32)[:, 3] = np.array(c["label"], np.uint32)
    h = pc.PointCloudHeader(c["fields"], [4, 4, 4, 4], [1, 1, 1, 1], Width=len(rec), Height=1)
    return pc.PointCloud(h, len(rec), rec)


def test_voxelgrid_table_golden(golden):
    g = golden("ref_voxelgrid.json")
    pp = _golden_cloud(g)
    for c in g["cases"]:
        opts = [voxelgrid.WithChunkSize(c["chunk"])] if any(c["chunk"]) else []
        out = voxelgrid.New(g["leaf"], *opts).Filter(pp)
        assert out.Points == len(c["expected"]) == out.PointCloudHeader.Width
        assert out.PointCloudHeader.Height == 1
        rec = out.Data.view(f32).reshape(-1, 4)
        assert np.array_equal(rec[:, :3], np.array(c["expected"], f32)), c["name"]
        assert rec.view(np.uint32)[:, 3].tolist() == c["expected_labels"], c["name"]


def _records(pts, stride, off, rng):
    n = len(pts)
    rec = rng.integers(0, 256, size=(n, stride), dtype=np.uint8)
    rec[:, off:off + 12] = pts.view(np.uint8).reshape(n, 12)
    return np.ascontiguousarray(rec)


@pytest.mark.parametrize("n,width,leaf,chunk,stride,off", [
    (1, 1.0, 0.1, (0, 0, 0), 12, 0),
    (2, 1.0, 0.1, (0, 0, 0), 12, 0),
    (1000, 1.6, 0.05, (0, 0, 0), 12, 0),
    (100000, 1.6, 0.05, (0, 0, 0), 12, 0),       # BASELINE config C1
    (100000, 1.6, 0.05, (0, 0, 0), 16, 0),       # x y z label
    (50000, 1.6, 0.05, (0, 0, 0), 20, 4),        # xyz not at offset 0
    (50000, 1.6, 0.05, (0, 0, 0), 15, 1),        # unaligned records (binaryFloat32Iterator path)
    (100000, 1.6, 0.05, (8, 8, 8), 12, 0),
    (100000, 1.6, 0.05, (8, 8, 1), 16, 0),
    (100000, 1.6, 0.05, (3, 5, 7), 12, 0),
    (100000, 1.6, 0.05, (64, 64, 64), 12, 0),    # clamped to one chunk
    (300000, 3.0, 0.02, (0, 0, 0), 12, 0),
    (300000, 3.0, 0.02, (64, 64, 64), 12, 0),
    (200000, 0.3, 0.02, (0, 0, 0), 12, 0),       # ~60 points per voxel: long sequential sums
])
def test_voxel_filter_vs_oracle(n, width, leaf, chunk, stride, off):
    rng = np.random.default_rng(n + stride)
    pts = synth.uniform_cloud(n, width, 40 + n % 97)
    rec = _records(pts, stride, off, rng)
    exp = O.voxel_filter(rec, n, stride, off, (leaf,) * 3, chunk)
    # drive the C ABI directly: stride/offset need not correspond to a field list
    import ctypes as C
    from pcgol_amd import _lib as L
    out = np.empty(n * stride, np.uint8)
    m = C.c_int64()
    leafv = np.full(3, leaf, f32)
    chunkv = np.asarray(chunk, np.int32)
    L.check(L.lib().pcgx_voxel_filter(L.ptr(rec), n, stride, off, L.ptr(leafv), L.ptr(chunkv), L.ptr(out),
                                      C.byref(m)))
    assert m.value * stride == len(exp)
    assert np.array_equal(out[: len(exp)], exp)


def test_voxel_offset_cloud_chunked():
    """vMin != 0: only the chunked path is defined there (non-chunked uses size = vMax)."""
    pts = synth.uniform_cloud(50000, 2.0, 9) - f32(0.7)
    exp = O.voxel_filter(pts, len(pts), 12, 0, (0.05, 0.04, 0.03), (16, 16, 16))
    out = voxelgrid.New((0.05, 0.04, 0.03), voxelgrid.WithChunkSize((16, 16, 16))).Filter(pts)
    assert np.array_equal(out.Data, exp)


def test_voxel_positive_offset_non_chunked():
    """vMin > 0, non-chunked: the size = vMax quirk (voxelgrid.go:46) over-allocates but is legal."""
    pts = synth.uniform_cloud(50000, 1.0, 11) + f32(0.5)
    exp = O.voxel_filter(pts, len(pts), 12, 0, (0.05,) * 3)
    out = voxelgrid.New((0.05,) * 3).Filter(pts)
    assert np.array_equal(out.Data, exp)


def test_voxel_errors():
    with pytest.raises(ErrNoPoint):
        voxelgrid.New((0.1,) * 3).Filter(np.zeros((0, 3), f32))
    # negative vMin in non-chunked mode: the Go code panics (index out of range); we report an error
    pts = synth.uniform_cloud(1000, 1.0, 5) - f32(10.0)
    with pytest.raises(O.OracleError):
        O.voxel_filter(pts, len(pts), 12, 0, (0.05,) * 3)
    with pytest.raises(PcgxError) as ei:
        voxelgrid.New((0.05,) * 3).Filter(pts)
    assert ei.value.code == 7


def test_minmax_golden_and_edges(golden):
    g = golden("ref_mat.json")["minmax"]
    mn, mx = pc.MinMaxVec3(np.array(g["points"], f32))
    assert np.array_equal(mn, np.array(g["expected_min"], f32))
    assert np.array_equal(mx, np.array(g["expected_max"], f32))
    with pytest.raises(ErrNoPoint):
        pc.MinMaxVec3(np.zeros((0, 3), f32))
    # -0 / +0 and NaN behave as in the sequential loop (minmax.go:13-23)
    pts = synth.uniform_cloud(100000, 2.0, 17) - f32(1.0)
    pts[777] = (np.nan, 0.0, -0.0)
    pts[5] = (-0.0, np.nan, 0.0)
    pts[60000:60010, 2] = 0.0
    omn, omx = O.minmax(pts, len(pts))
    mn, mx = pc.MinMaxVec3(pts)
    assert np.array_equal(mn.view(np.uint32), omn.view(np.uint32))
    assert np.array_equal(mx.view(np.uint32), omx.view(np.uint32))
    pts[0, 1] = np.nan  # NaN at index 0 sticks
    omn, omx = O.minmax(pts, len(pts))
    mn, mx = pc.MinMaxVec3(pts)
    assert np.array_equal(mn.view(np.uint32), omn.view(np.uint32))
    assert np.array_equal(mx.view(np.uint32), omx.view(np.uint32))


def test_c3_scale_properties():
    """BASELINE config C3 (10M, leaf 0.02) at full size through size-independent properties:
    idempotence is NOT a property of this filter (centroids move), so check instead
    (a) M == number of distinct voxel keys, (b) every output point lies in its voxel's
    cell (up to rounding), (c) outputs are ordered by ascending voxel key, and
    (d) a 1M prefix is byte-identical to the oracle."""
    c = synth.c3_voxel()
    pts = c["points"]
    leaf = f32(0.02)
    out = voxelgrid.New(c["leaf"]).Filter(pts).Vec3()
    vmin = pts.min(axis=0)
    vmax = pts.max(axis=0)
    xs, ys = int(f32(vmax[0]) / leaf), int(f32(vmax[1]) / leaf)

    def keys(p):
        q = ((p - vmin) / leaf).astype(np.int64)
        return q[:, 0] + xs * (q[:, 1] + ys * q[:, 2])

    kin = keys(pts)
    assert len(out) == len(np.unique(kin))
    # Outputs come in ascending key order, one per distinct key.  Cells with x == xs alias
    # (x = 0, y + 1) through the reference's stride quirk (voxelgrid.go:137-138,151), so their
    # merged centroid lies in neither cell (same for y == ys): check the un-aliased keys only;
    # a centroid may still round across a cell face by an ulp, hence the tolerance.
    kout = keys(out)
    uk = np.unique(kin)
    plain = ((uk % xs) != 0) & (((uk // xs) % ys) != 0)
    assert np.mean(kout[plain] == uk[plain]) > 0.9999
    sub = np.ascontiguousarray(pts[:1_000_000])
    exp = O.voxel_filter(sub, len(sub), 12, 0, c["leaf"])
    got = voxelgrid.New(c["leaf"]).Filter(sub)
    assert np.array_equal(got.Data, exp)


@pytest.mark.parametrize("chunk", [None, (64, 64, 64)])
def test_c3_full_size_against_the_oracle(chunk):
    """BASELINE config C3 (10M points, leaf 0.02, cube 3 m) at FULL size, byte for byte against the
    oracle's dense-array restatement (voxelgrid.go:35-187; its 151^3 x 32 B array is 110 MB), in the
    plain mode and with WithChunkSize{64,64,64}."""
    c = synth.c3_voxel()
    pts = c["points"]
    if chunk is None:
        exp = O.voxel_filter(pts, len(pts), 12, 0, c["leaf"])
        got = voxelgrid.New(c["leaf"]).Filter(pts)
    else:
        exp = O.voxel_filter(pts, len(pts), 12, 0, c["leaf"], chunk)
        got = voxelgrid.New(c["leaf"], voxelgrid.WithChunkSize(chunk)).Filter(pts)
    assert got.Data.shape == exp.shape and 3_000_000 < len(exp) // 12 < 3_400_000
    assert np.array_equal(got.Data, exp)


def test_voxel_filter_more_shapes_vs_oracle():
    """Larger clouds than the table above, against the oracle's bytes: plain and chunked, a record
    with extra fields around xyz (the first point's whole record is carried over), ~400 points per
    voxel (long sequential sums) and a cloud squeezed into a thin slab (most of the key range empty)."""
    import ctypes as C
    from pcgol_amd import _lib as L
    rng = np.random.Generator(np.random.PCG64(3))
    cases = []
    pts = synth.uniform_cloud(400_000, 3.0, 5)
    cases.append((pts, 12, 0, (0.02, 0.02, 0.02), (0, 0, 0)))
    cases.append((pts, 12, 0, (0.03, 0.02, 0.025), (16, 8, 32)))
    rec = np.zeros((len(pts), 5), np.float32)        # label | x y z | intensity
    rec[:, 1:4] = pts
    rec[:, 0] = np.arange(len(pts))
    rec[:, 4] = rng.random(len(pts))
    cases.append((np.ascontiguousarray(rec), 20, 4, (0.02, 0.02, 0.02), (0, 0, 0)))
    cases.append((pts, 12, 0, (0.3, 0.3, 0.3), (0, 0, 0)))
    slab = pts.copy()
    slab[:, 2] = slab[:, 2] * f32(0.001)
    cases.append((np.ascontiguousarray(slab), 12, 0, (0.004, 0.004, 0.004), (0, 0, 0)))
    for data, stride, off, leaf, chunk in cases:
        n = len(data)
        exp = O.voxel_filter(data, n, stride, off, leaf, chunk)
        out = np.empty(n * stride, np.uint8)
        m = C.c_int64()
        leafv, chunkv = np.asarray(leaf, f32), np.asarray(chunk, np.int32)   # kept alive across the call
        L.check(L.lib().pcgx_voxel_filter(L.ptr(data), n, stride, off, L.ptr(leafv), L.ptr(chunkv), L.ptr(out),
                                          C.byref(m)))
        assert m.value * stride == len(exp) and np.array_equal(out[: len(exp)], exp), (stride, leaf, chunk)


def test_chunked_two_sort_path_still_matches(monkeypatch):
    """Chunked mode normally sorts ONE combined (chunk id, cell) key; grids whose two indices do not
    fit 32 bits fall back to two stable sorts.  Force that path and compare both with the oracle."""
    pts = synth.uniform_cloud(60000, 3.0, 77) - f32(0.7)
    leaf, chunk = (0.05, 0.04, 0.06), (7, 5, 9)
    exp = O.voxel_filter(pts, len(pts), 12, 0, leaf, chunk)
    got = voxelgrid.New(leaf, voxelgrid.WithChunkSize(chunk)).Filter(pts)
    assert np.array_equal(got.Data, exp)
    monkeypatch.setenv("PCGX_VOXEL_TWO_SORTS", "1")
    got2 = voxelgrid.New(leaf, voxelgrid.WithChunkSize(chunk)).Filter(pts)
    assert np.array_equal(got2.Data, exp)
