"""The C++ host mirror (pcgol_amd/host/pcgx.hpp) over the C ABI: compiled with g++ everywhere
(CPU check: it builds and links against libpcgx.so), run on the GPU box against the
reference's known-answer tables (tests/golden/ref_*.json)."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "host_mirror.cpp")


def _build(tmpdir):
    from pcgol_amd import build as B
    B.build()
    exe = os.path.join(str(tmpdir), "host_mirror")
    libdir = os.path.join(ROOT, "pcgol_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-o", exe, SRC, "-L" + libdir, "-lpcgx",
                           "-Wl,-rpath," + libdir])
    return exe


def O_segment_order(sg):
    import oracle as O
    og = O.BucketGrid(sg["resolution"], sg["size"], np.array(sg["origin"], np.float32))
    og.add_all(np.array(sg["points"], np.float32))
    return og.segment(sg["seed_point"]).tolist()


def test_cpp_host_mirror_compiles_and_links(tmp_path):
    exe = _build(tmp_path)
    assert os.path.getsize(exe) > 0
    out = subprocess.run(["ldd", exe], capture_output=True, text=True).stdout
    assert "libpcgx.so" in out and "not found" not in out.split("libpcgx.so")[1].split("\n")[0]


@pytest.mark.gpu
def test_cpp_host_mirror_known_answers(tmp_path, golden):
    exe = _build(tmp_path)
    kd = golden("ref_kdtree.json")
    vx = golden("ref_voxelgrid.json")
    lines = []
    pts = kd["test_cloud"]["points"]
    lines.append("P %d" % len(pts))
    lines += ["%r %r %r" % tuple(map(float, p)) for p in pts]
    ncases = kd["nearest"]["cases"]
    for md in kd["nearest"]["min_dist"]:
        lines.append("M %r" % md)
        for c in ncases:  # max_range differs per case: one batch each
            lines.append("Q 1 %r" % c["max_range"])
            lines.append("%r %r %r" % tuple(map(float, c["p"])))
    lines.append("M 0")
    lines.append("I 0.01 -0.02 0.015 2.0 3")
    rp = kd["range"]["points"]
    lines.append("P %d" % len(rp))
    lines += ["%r %r %r" % tuple(map(float, p)) for p in rp]
    lines.append("R %d" % len(kd["range"]["cases"]))
    lines += ["%r %r %r %r" % (*map(float, c["p"]), c["max_range"]) for c in kd["range"]["cases"]]
    sg = golden("ref_segment.json")["flood_fill"]
    lines.append("P %d" % len(sg["points"]))
    lines += ["%r %r %r" % tuple(map(float, p)) for p in sg["points"]]
    o = float(np.float32(sg["origin"][0]))
    lines.append("G %r %d %d %d %r %r %r %r %r %r" % (sg["resolution"], *sg["size"], o, o, o, *sg["seed_point"]))
    lines.append("W 1 0.5 0.5 0.5 0.051")
    lines.append("V %d" % len(vx["cloud"]["xyz"]))
    lines += ["%r %r %r %d" % (*map(float, p), l) for p, l in zip(vx["cloud"]["xyz"], vx["cloud"]["label"])]
    for c in vx["cases"]:
        lines.append("L %r %r %r %d %d %d" % (*vx["leaf"], *c["chunk"]))
    inp = tmp_path / "in.txt"
    inp.write_text("\n".join(lines) + "\n")
    r = subprocess.run([exe, str(inp)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    out = r.stdout.strip().split("\n")
    assert out[0] == "tree len 7 depth %d" % kd["max_depth"]["cases"][-1]["expected"]
    near = [l.split() for l in out if l.startswith("nearest")]
    assert len(near) == 2 * len(ncases)
    eps = kd["nearest"]["eps"]
    for k, l in enumerate(near):
        c = ncases[k % len(ncases)]
        assert int(l[1]) == c["id"] and abs(float(l[2]) - c["dist_sq"]) <= eps, (k, l, c)
    rng = [l for l in out if l.startswith("range")]
    for l, c in zip(rng, kd["range"]["cases"]):
        got = [(int(t.split(":")[0]), float(t.split(":")[1])) for t in l.split()[1:]]
        assert [g[0] for g in got] == [nb[0] for nb in c["neighbors"]]
        assert all(abs(g[1] - nb[1]) <= kd["range"]["eps"] for g, nb in zip(got, c["neighbors"]))
    vox = [l for l in out if l.startswith("voxel")]
    for l, c in zip(vox, vx["cases"]):
        recs = [t.split(",") for t in l.split()[1:]]
        assert [int(r[3]) for r in recs] == c["expected_labels"], c["name"]
        got = np.array([[float(v) for v in r[:3]] for r in recs], np.float32)
        assert np.array_equal(got, np.array(c["expected"], np.float32)), c["name"]
    seg = [l for l in out if l.startswith("segment")][0].split()[1:]
    assert sorted(int(v) for v in seg) == sg["expected_sorted"]
    assert [int(v) for v in seg] == O_segment_order(sg)
    import oracle as O
    spts = np.array(sg["points"], np.float32)
    og = O.BucketGrid(sg["resolution"], sg["size"], np.array(sg["origin"], np.float32))
    og.add_all(spts)
    assert [l for l in out if l.startswith("get ")][0] == "get %d len %d" % (len(og.get(sg["seed_point"])), 64 ** 3)
    # region growing through the C++ mirror == the oracle's BFS (one property value, maxRange 0.051)
    # (the mirror's Segment returns the reference's own BFS order)
    exp = O.region_growing_segment(O.KDTree(spts), np.zeros(len(spts), np.uint32), [0.5, 0.5, 0.5], 0.051).tolist()
    assert [int(v) for v in [l for l in out if l.startswith("region")][0].split()[1:]] == exp and len(exp) >= 3
    icp = [l for l in out if l.startswith("icp ")][0].split()
    # same Fit through the Python mirror (same C ABI): identical transform
    from pcgol_amd import icp as picp, kdtree
    base = np.array(pts, np.float32)
    target = base + np.array([0.01, -0.02, 0.015], np.float32)
    tr, st = picp.PointToPointICPGradient(picp.PointToPointEvaluator(picp.NearestPointCorresponder(2.0), 3)).Fit(
        kdtree.New(base), target)
    assert int(icp[2]) == st.NumIteration
    assert np.array_equal(np.array([float(v) for v in icp[6:22]], np.float32), tr)
    assert "icp_minpairs ErrNotEnoughPairs" in out and "empty ErrNoPoint" in out
    # the float64-tree mode through both mirrors
    tr64, st64 = picp.PointToPointICPGradient(picp.PointToPointEvaluator(
        picp.NearestPointCorresponder(2.0), 3, SumsMode=picp.SumsF64Tree)).Fit(kdtree.New(base), target)
    f64 = [l for l in out if l.startswith("icp_f64 ")][0].split()
    assert int(f64[2]) == st64.NumIteration
    assert np.array_equal(np.array([float(v) for v in f64[6:22]], np.float32), tr64)
    # default (reference) sums + a built-in weight through the C++ mirror == through a Python session, bit for bit
    s = picp.IcpSession(kdtree.New(base), target, 2.0, 3, None, None, 0, WeightFn=picp.WeightHuber(0.0004))
    for _ in range(20):
        s.step()
    tr_s, st_s, _ = s.result()
    s.close()
    strict = [l for l in out if l.startswith("icp_strict ")][0].split()
    assert int(strict[2]) == st_s.NumIteration
    assert np.array_equal(np.array([float(v) for v in strict[6:22]], np.float32), tr_s)
    assert "sharded1_icp same 1" in out and "sharded1_voxel same 1 world 1" in out
    # the reference's package surface name for name (pcgx::kdtree::New, pcgx::voxelgrid::New + WithChunkSize,
    # pcgx::icp::PointToPointICPGradient{Evaluator, UpdaterFactory}.Fit): the same transform, the same records
    nl = [l for l in out if l.startswith("named_icp ")][0].split()
    assert nl[2] == "1" and nl[4] == "1" and float(nl[6]) > 0.0
    assert any(l.startswith("named_voxel same 1 records ") and int(l.split()[-1]) >= 1 for l in out)
    # the Go seams through the C++ mirror: With() shares the device tree and carries the option; the corresponder's
    # pairs are the oracle's (correspondence.go:22-37), in target order
    wl = [l for l in out if l.startswith("with ")][0].split()
    assert float(wl[2]) == 0.25 and wl[4] == "1" and int(wl[6]) == len(base)
    pl = [l for l in out if l.startswith("pairs ")][0].split()
    ob, ot, od = O.icp_pairs(O.KDTree(base), target, 2.0)
    assert int(pl[1]) == len(ob)
    for k, tok in enumerate(pl[2:]):
        b, t, d = tok.split(":")
        assert int(b) == ob[k] and int(t) == ot[k] and np.float32(float(d)) == od[k]
