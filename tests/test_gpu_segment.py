"""GPU parity of the bucket voxel grid (pc/storage/voxelgrid), voxel flood fill
(pc/segmentation/voxelgrid) and region growing (pc/segmentation/regiongrowing) through the C ABI:
the reference's known-answer tables (tests/golden/ref_segment.json) and the oracle on seeded
random inputs.  Buckets are compared exactly (insertion order inside a voxel); Segment results as
sets, like the reference's own tests (its BFS discovery order is not reproduced, include/pcgx.h)."""
import numpy as np
import pytest

import oracle as O
from pcgol_amd import kdtree, segmentation, synth
from segment_scene import region_growing_scene

pytestmark = pytest.mark.gpu
f32 = np.float32


def test_bucket_grid_table(golden):
    g = golden("ref_segment.json")["bucket_grid"]
    pts = np.array(g["points"], f32)
    v = segmentation.StorageVoxelGrid(g["resolution"], g["size"], g["origin"])
    assert v.AddAll(pts).tolist() == g["add_ok"]
    for p, exp in zip(pts, g["get"]):
        got = v.Get(p)
        assert (got is None) if exp is None else (got.tolist() == exp)
    assert v.Indice().tolist() == [1, 2, 3] and v.Len() == 64 ** 3
    mn, mx = v.MinMax()
    assert np.array_equal(mn, np.array(g["origin"], f32)) and np.allclose(mx - mn, 64 * 0.05)


def test_flood_fill_table(golden):
    g = golden("ref_segment.json")["flood_fill"]
    v = segmentation.SegmentationVoxelGrid(g["resolution"], g["size"], np.array(g["origin"], f32))
    v.AddAll(np.array(g["points"], f32))
    assert sorted(v.Segment(g["seed_point"]).tolist()) == g["expected_sorted"]
    assert sorted(v.Segment(g["seed_point"], order="address").tolist()) == g["expected_sorted"]
    o = O.BucketGrid(g["resolution"], g["size"], np.array(g["origin"], f32))
    o.add_all(np.array(g["points"], f32))
    assert v.Segment(g["seed_point"]).tolist() == o.segment(g["seed_point"]).tolist()  # the reference's order
    assert v.Segment([0.3, 0.3, 0.3]).tolist() == [] and v.Segment([9, 9, 9]).tolist() == []
    assert v.Segment([0.3, 0.3, 0.3], order="address").tolist() == []


@pytest.mark.parametrize("n,res,seed", [(3000, 0.25, 0), (200_000, 0.08, 1)])
def test_bucket_grid_and_flood_fill_vs_oracle(n, res, seed):
    """Random cloud partly outside the grid, with a label field (stride 16): buckets identical to
    the oracle's (order included), every component identical to the oracle's flood fill."""
    pts = synth.uniform_cloud(n, 10.0, 60 + seed) - f32(1.0)
    rec = np.zeros((n, 4), f32)
    rec[:, :3] = pts
    from pcgol_amd import pc
    cloud = pc.PointCloud(pc.PointCloudHeader(["x", "y", "z", "label"], [4] * 4, [1] * 4), n, rec.view(np.uint8).reshape(-1))
    size, origin = [90, 80, 100], [0.5, -0.25, 0.125]
    v = segmentation.SegmentationVoxelGrid(res, size, origin)
    added = v.AddAll(cloud)
    o = O.BucketGrid(res, size, origin)
    oadded = np.array([o.add(p, i) for i, p in enumerate(pts)])
    assert np.array_equal(added, oadded) and 0 < added.sum() < n
    assert np.array_equal(v.Indice(), o.indice())
    rng = np.random.default_rng(seed)
    for i in rng.choice(n, 200, replace=False):
        a, ok = v.Addr(pts[i])
        assert (a, ok) == o.addr(pts[i])
        got, exp = v.Get(pts[i]), o.get(pts[i])
        assert (got is None and exp is None) or np.array_equal(got, exp)
    comp = v.Components()
    assert np.array_equal(comp >= 0, added)
    addrs = v.PointAddrs()
    seen = set()
    for i in rng.choice(np.nonzero(added)[0], 40, replace=False):
        if comp[i] in seen:
            continue
        seen.add(int(comp[i]))
        exact = o.segment(pts[i])
        assert np.array_equal(v.Segment(pts[i]), exact)  # id for id in the reference's BFS order
        exp = np.sort(exact)
        assert np.array_equal(np.sort(v.Segment(pts[i], order="address")), exp)
        assert np.array_equal(np.nonzero(comp == comp[i])[0], exp)
        assert comp[i] == addrs[exp].min()  # canonical id: smallest voxel address of the component


@pytest.mark.parametrize("seed", [0, 1])
def test_region_growing_table(golden, seed):
    g = golden("ref_segment.json")["region_growing"]
    pts, labels, ids = region_growing_scene(g, seed)
    t = kdtree.New(pts)
    rg = segmentation.RegionGrowing.New(t, labels)
    ot = O.KDTree(pts)
    for c in g["cases"]:
        exp = sorted(sum((ids[o] for o in c["objects"]), []))
        got = rg.Segment(c["p"], c["max_range"], order="id")
        assert got.tolist() == exp, c["name"]  # ascending id
        oseg = O.region_growing_segment(ot, labels, c["p"], c["max_range"])
        assert sorted(oseg.tolist()) == exp
        assert rg.Segment(c["p"], c["max_range"]).tolist() == oseg.tolist(), c["name"]  # the reference's order
    assert rg.Segment([50, 50, 50], 0.1).tolist() == [] and rg.Segment([50, 50, 50], 0.1, order="id").tolist() == []


def test_region_growing_vs_oracle_random():
    """Random cloud, random property values: every seed's region equals the oracle's BFS."""
    n = 20000
    pts = synth.uniform_cloud(n, 4.0, 70)
    rng = np.random.default_rng(7)
    labels = rng.integers(0, 3, n).astype(np.uint32)
    t, ot = kdtree.New(pts), O.KDTree(pts)
    rg = segmentation.RegionGrowing(t, labels)
    for mr in (0.12, 0.2):
        comp = rg.Components(mr)
        assert np.all(comp <= np.arange(n)) and np.array_equal(labels[comp], labels)
        for k in range(25):
            p = pts[rng.integers(n)] + f32(0.01)
            exact = O.region_growing_segment(ot, labels, p, mr)
            assert np.array_equal(rg.Segment(p, mr, order="id"), np.sort(exact)), (mr, k)
            if k < 8:
                assert np.array_equal(rg.Segment(p, mr), exact), (mr, k)  # id for id, BFS order
    # after DeletePoint the regions are those of the remaining points
    gone = rng.choice(n, n // 4, replace=False)
    t.DeletePoints(gone)
    for i in gone:
        ot.delete_point(int(i))
    rg2 = segmentation.RegionGrowing(t, labels)
    for k in range(10):
        p = pts[rng.integers(n)] + f32(0.01)
        exact = O.region_growing_segment(ot, labels, p, 0.2)
        assert np.array_equal(rg2.Segment(p, 0.2, order="id"), np.sort(exact))
        if k < 3:
            assert np.array_equal(rg2.Segment(p, 0.2), exact)


@pytest.mark.parametrize("kind", ["uniform", "lattice", "sheet"])
def test_region_growing_on_the_grid_equals_the_walk(kind, monkeypatch):
    """Components(maxRange) joins every point with its Range() neighbours of the same property value
    (regiongrowing.go:43-47).  On a handle with a uniform grid the neighbourhoods come out of the grid's cells and
    the union-find works in cell order (csrc/segment.hip); PCGX_RANGE_WALK=1 keeps the walk.  Same component for
    every point -- named by its smallest id either way."""
    rng = np.random.default_rng(5)
    n = 150_000
    if kind == "uniform":
        pts, r = synth.uniform_cloud(n, 5.0, 31), 0.12
    elif kind == "lattice":  # repeated sites, distances tied at the bound's doorstep
        pts, r = rng.integers(0, 40, size=(n, 3)).astype(np.float32), 1.0001
    else:
        pts = np.stack([rng.uniform(0, 8, n), rng.uniform(0, 8, n), rng.normal(0, 0.005, n)], axis=1).astype(np.float32)
        r = 0.05
    labels = rng.integers(0, 3, n).astype(np.uint32)
    t = kdtree.New(pts)
    monkeypatch.delenv("PCGX_RANGE_WALK", raising=False)
    a = segmentation.RegionGrowing(t, labels).Components(r)
    monkeypatch.setenv("PCGX_RANGE_WALK", "1")
    b = segmentation.RegionGrowing(t, labels).Components(r)
    monkeypatch.delenv("PCGX_RANGE_WALK", raising=False)
    assert np.array_equal(a, b)
    assert np.all(a <= np.arange(n)) and np.array_equal(a[a], a)   # a component is named by its smallest member
    assert np.array_equal(labels[a], labels)
