"""Pins the CPU oracle (oracle/pcgol_oracle.c) against the reference's own
known-answer tables (tests/golden/ref_*.json, transcribed from the Go tests).
CPU only."""
import numpy as np
import pytest

import oracle as O

f32 = np.float32


def _tree_from_dump(dump):
    def rec(k):
        if k < 0:
            return None
        i, d, c0, c1 = (int(v) for v in dump[k])
        return [i, d, rec(c0), rec(c1)]
    return rec(0)


def test_kdtree_shape(golden):
    g = golden("ref_kdtree.json")
    t = O.KDTree(g["test_cloud"]["points"])
    assert _tree_from_dump(t.dump()) == g["expected_tree"]["root"]
    # in-order id list == final state of the reference's in-place sorted slice
    assert t.inorder().tolist() == [5, 4, 1, 3, 2, 0, 6]


def test_kdtree_max_depth(golden):
    for c in golden("ref_kdtree.json")["max_depth"]["cases"]:
        assert O.KDTree(c["points"]).max_depth() == c["expected"]


def test_kdtree_nearest_table(golden):
    g = golden("ref_kdtree.json")
    eps = g["nearest"]["eps"]
    for md in g["nearest"]["min_dist"]:
        t = O.KDTree(g["test_cloud"]["points"], min_dist_sq=f32(md) * f32(md))
        for c in g["nearest"]["cases"]:
            i, d = t.nearest(c["p"], c["max_range"])
            assert i == c["id"], (md, c)
            assert abs(float(d) - c["dist_sq"]) <= eps, (md, c, d)


def test_kdtree_search_leaf(golden):
    g = golden("ref_kdtree.json")
    t = O.KDTree(g["test_cloud"]["points"])
    for c in g["search_leaf"]["cases"]:
        assert t.search_leaf(c["p"]) == c["id"]


def test_kdtree_range_table(golden):
    g = golden("ref_kdtree.json")["range"]
    t = O.KDTree(g["points"])
    for c in g["cases"]:
        ids, dsq = t.range(c["p"], c["max_range"])
        assert ids.tolist() == [n[0] for n in c["neighbors"]]
        for d, n in zip(dsq, c["neighbors"]):
            assert abs(float(d) - n[1]) <= g["eps"]


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_kdtree_nearest_equals_naive(golden, seed):
    """kdtree_test.go:794-834: ID and DistSq must == brute force."""
    g = golden("ref_kdtree.json")["random_property"]
    rng = np.random.default_rng(seed)
    w = f32(g["width"])
    pts = rng.random((g["n_points"], 3), dtype=f32) * w
    t = O.KDTree(pts)
    for _ in range(g["n_queries"]):
        p = rng.random(3, dtype=f32) * w
        mr = float(rng.random(dtype=f32) * w)
        assert t.nearest(p, mr) == O.naive_nearest(pts, p, mr)


def test_kdtree_range_equals_naive():
    """kdtree_test.go:887-924."""
    rng = np.random.default_rng(5)
    pts = rng.random((100, 3), dtype=f32) * f32(10)
    t = O.KDTree(pts)
    for _ in range(100):
        p = rng.random(3, dtype=f32) * f32(10)
        mr = f32(rng.random(dtype=f32) * f32(10))
        ids, dsq = t.range(p, float(mr))
        assert np.all(np.diff(dsq) >= 0)
        d = pts - p
        dn = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
        sel = np.nonzero(dn < mr * mr)[0]
        assert sorted(zip(dsq.tolist(), ids.tolist())) == sorted(zip(dn[sel].tolist(), sel.tolist()))


def test_kdtree_find_minimum_table(golden):
    """kdtree_test.go:388-411 + :836-862 (vs brute force on random clouds)."""
    g = golden("ref_kdtree.json")
    t = O.KDTree(np.array(g["test_cloud"]["points"], f32))
    for c in g["find_minimum"]["cases"]:
        if c.get("error"):
            with pytest.raises(ValueError):
                t.find_minimum(c["dim"])
        else:
            assert t.find_minimum(c["dim"]) == c["id"]
    rng = np.random.default_rng(11)
    for _ in range(100):
        pts = rng.random((100, 3), dtype=f32) * f32(10)
        tr = O.KDTree(pts)
        for dim in range(3):
            assert tr.find_minimum(dim) == int(np.argmin(pts[:, dim]))  # first strict minimum


def test_kdtree_delete_point_tables(golden):
    """kdtree_test.go:413-729: exact trees after each deletion of every named sequence."""
    g = golden("ref_kdtree.json")
    pts = np.array(g["test_cloud"]["points"], f32)
    for name, steps in g["delete_point"]["sequences"].items():
        t = O.KDTree(pts)
        for st in steps:
            if st["has_error"]:
                with pytest.raises(IndexError):
                    t.delete_point(st["pid"])
            else:
                t.delete_point(st["pid"])
            assert t.tree() == st["tree"], (name, st["pid"])


def test_kdtree_delete_on_line_and_everything(golden):
    """kdtree_test.go:731-751, and root == nil after the last deletion (kdtree.go:84-86,150-152)."""
    g = golden("ref_kdtree.json")["delete_on_line"]
    pts = np.array(g["points"], f32)
    t = O.KDTree(pts)
    for i in range(len(pts)):
        t.delete_point(i)
        assert t.nearest(pts[i], g["max_range"])[0] < 0
    assert t.tree() is None
    assert t.nearest(pts[0], 100.0) == (-1, f32(100.0) * f32(100.0))
    assert len(t.range(pts[0], 100.0)[0]) == 0


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_kdtree_delete_then_nearest_equals_naive(seed):
    """kdtree_test.go:864-885: delete a third of the points, then ID and DistSq must == brute
    force over the remaining points."""
    rng = np.random.default_rng(100 + seed)
    pts = rng.random((100, 3), dtype=f32) * f32(10)
    t = O.KDTree(pts)
    gone = rng.permutation(33)
    for i in gone:
        t.delete_point(int(i))
    keep = np.setdiff1d(np.arange(100), gone)
    for _ in range(100):
        p = rng.random(3, dtype=f32) * f32(10)
        mr = float(rng.random(dtype=f32) * f32(10))
        i, d = O.naive_nearest(pts[keep], p, mr)
        assert t.nearest(p, mr) == ((int(keep[i]) if i >= 0 else -1), d)


def _voxel_cloud(g):
    c = g["cloud"]
    rec = np.zeros((len(c["xyz"]), 4), np.float32)
    rec[:, :3] = np.array(c["xyz"], np.float32)
    rec.view(np.uint32)[:, 3] = np.array(c["label"], np.uint32)
    return rec


def test_voxelgrid_table(golden):
    g = golden("ref_voxelgrid.json")
    rec = _voxel_cloud(g)
    for c in g["cases"]:
        out = O.voxel_filter(rec, len(rec), 16, 0, g["leaf"], c["chunk"]).view(np.float32).reshape(-1, 4)
        exp = np.array(c["expected"], np.float32)
        assert out.shape[0] == len(exp), c["name"]
        assert np.array_equal(out[:, :3], exp), (c["name"], out[:, :3])  # Vec3.Equal: exact
        assert out.view(np.uint32)[:, 3].tolist() == c["expected_labels"], c["name"]


def test_minmax(golden):
    g = golden("ref_mat.json")["minmax"]
    pts = np.array(g["points"], np.float32)
    mn, mx = O.minmax(pts, len(pts))
    assert np.array_equal(mn, np.array(g["expected_min"], np.float32))
    assert np.array_equal(mx, np.array(g["expected_max"], np.float32))
    with pytest.raises(O.OracleError):
        O.minmax(np.zeros(0, np.float32), 0)


def test_translate_rotate(golden):
    g = golden("ref_mat.json")
    tr = g["translate"]
    assert O.translate(*tr["elements"]["args"]).tolist() == tr["elements"]["expected"]
    m = O.translate(*tr["example"]["args"])
    assert O.mat4_transform(m, [tr["example"]["v"]])[0].tolist() == tr["example"]["expected"]
    for c in g["rotate"]["cases"]:
        ang = f32(np.float64(np.pi) * c["ang_pi"])  # Go: untyped const pi/2 -> float32
        m = O.rotate(*c["axis"], float(ang))
        assert np.all(np.abs(m - np.array(c["expected"], np.float32)) < g["rotate"]["tolerance"]), c["name"]


def test_transform_vs_naive(golden):
    g = golden("ref_mat.json")["transform_vs_naive"]
    sc = np.diag(np.array(g["scale"] + [1.0], np.float32)).reshape(-1)
    m = O.translate(*g["translate"])
    for f in (sc, O.rotate(1, 0, 0, g["rot_ang"]), O.rotate(0, 1, 0, g["rot_ang"]), O.rotate(0, 0, 1, g["rot_ang"])):
        m = O.mat4_mul(m, f)
    v = np.array(g["v"], np.float32)
    out = O.mat4_transform(m, [v])[0]
    M = m.reshape(4, 4).T.astype(np.float64)  # column-major storage
    naive = (M @ np.array([*v, 1.0]))[:3]
    assert np.all(np.abs(out - naive) < g["tolerance"])


def test_icp_corresponder(golden):
    g = golden("ref_icp.json")["corresponder"]
    t = O.KDTree(g["base"])
    b, tid, d = O.icp_pairs(t, g["targets"], g["max_dist"])
    got = [[int(x), int(y), float(z)] for x, y, z in zip(b, tid, d)]
    assert got == g["expected_pairs"]


def test_icp_evaluator(golden):
    g = golden("ref_icp.json")["evaluator"]
    base = np.array(g["base"], np.float32)
    delta = np.array(g["delta"], np.float32)
    target = base[g["target_base_ids"]] + delta
    t = O.KDTree(base)
    ev = O.icp_evaluate(t, target, g["max_dist"], g["min_pairs"])
    assert ev["value"] == f32(g["expected_value"])  # exact (evaluator_test.go:40-42)
    assert ev["npairs"] == 3
    fct = f32(g["step_factor"])
    dR = O.rodrigues(ev["gradient"][3:] * fct)
    ev2 = O.icp_evaluate(t, O.mat4_transform(dR, target), g["max_dist"], g["min_pairs"])
    assert ev2["value"] < ev["value"]
    ev3 = O.icp_evaluate(t, target + ev["gradient"][:3] * fct, g["max_dist"], g["min_pairs"])
    assert ev3["value"] < ev["value"]
    with pytest.raises(O.OracleError) as ei:
        O.icp_evaluate(t, target, g["max_dist"], 4)
    assert ei.value.code == O.ORC_E_NOT_ENOUGH_PAIRS


def _delta(ops):
    m = None
    for op in ops:
        f = O.translate(*op[1:]) if op[0] == "trans" else O.rotate(*op[1:])
        m = f if m is None else O.mat4_mul(m, f)
    return m


def test_icp_fit_poses(golden):
    g = golden("ref_icp.json")["fit"]
    idx = g["indices"]
    for name, base in g["bases"].items():
        base = np.array(base, np.float32)
        for ops in g["deltas"]:
            target = O.mat4_transform(_delta(ops), base[idx])
            t = O.KDTree(base, min_dist_sq=g["min_dist_sq"])
            r = O.icp_fit(t, target, g["max_dist"], g["min_pairs"])
            moved = O.mat4_transform(r["trans"], target)
            d = moved - base[idx]
            res = f32(0)
            for row in d:
                res = f32(res + f32(f32(f32(row[0] * row[0]) + f32(row[1] * row[1])) + f32(row[2] * row[2])))
            res = f32(res / f32(len(idx)))
            assert res <= g["max_residual"], (name, ops, res)
            assert 1 <= r["num_iteration"] <= 20


def test_rodrigues_sweep(golden):
    g = golden("ref_icp.json")["rodrigues"]
    vals = []
    v = f32(g["start"])
    while v < g["stop"]:
        vals.append(v)
        v = f32(v + f32(g["step"]))
    vals = vals[::7]  # subsample the 100^3 sweep (CPU time); endpoints of the table included
    for vx in vals:
        for vy in vals:
            for vz in vals:
                vec = np.array([vx, vy, vz], np.float32)
                r = O.rodrigues(vec)
                nsq = f32(f32(f32(vx * vx) + f32(vy * vy)) + f32(vz * vz))
                norm = f32(np.sqrt(np.float64(nsq)))
                vn = vec * f32(f32(1.0) / norm)
                e = O.rotate(float(vn[0]), float(vn[1]), float(vn[2]), float(norm))
                assert np.all(np.abs(r - e) <= g["eps"]), (vec, r, e)


# ------------------------------------------------ bucket grid / flood fill / region growing

def test_bucket_grid_table(golden):
    """pc/storage/voxelgrid/voxelgrid_test.go:10-87: Add and AddByAddr."""
    g = golden("ref_segment.json")["bucket_grid"]
    pts = np.array(g["points"], f32)
    for mode in ("add", "add_by_addr"):
        v = O.BucketGrid(g["resolution"], g["size"], g["origin"])
        for i, p in enumerate(pts):
            if mode == "add":
                assert v.add(p, i) == g["add_ok"][i]
            else:
                a, ok = v.addr(p)
                assert ok == g["add_ok"][i]
                if ok:
                    v.add_by_addr(a, i)
        for p, exp in zip(pts, g["get"]):
            got = v.get(p)
            assert (got is None) if exp is None else (got.tolist() == exp)
        assert v.indice().tolist() == [1, 2, 3]


def test_flood_fill_table(golden):
    """pc/segmentation/voxelgrid/voxelgrid_test.go:11-40."""
    g = golden("ref_segment.json")["flood_fill"]
    v = O.BucketGrid(g["resolution"], g["size"], np.array(g["origin"], f32))
    v.add_all(np.array(g["points"], f32))
    assert sorted(v.segment(g["seed_point"]).tolist()) == g["expected_sorted"]
    assert v.segment([0.3, 0.3, 0.3]).tolist() == []   # empty seed voxel: nothing (voxelgrid.go:57-60)
    assert v.segment([9, 9, 9]).tolist() == []         # seed outside the grid (:41-44)


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_region_growing_table(golden, seed):
    """pc/segmentation/regiongrowing/regiongrowing_test.go:15-175."""
    from segment_scene import region_growing_scene
    g = golden("ref_segment.json")["region_growing"]
    pts, labels, ids = region_growing_scene(g, seed)
    t = O.KDTree(pts)
    for c in g["cases"]:
        exp = sorted(sum((ids[o] for o in c["objects"]), []))
        got = O.region_growing_segment(t, labels, c["p"], c["max_range"])
        assert sorted(got.tolist()) == exp, c["name"]
    assert O.region_growing_segment(t, labels, [50, 50, 50], 0.1).tolist() == []
