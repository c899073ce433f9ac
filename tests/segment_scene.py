"""Builds the region-growing test scene of the reference (regiongrowing_test.go:15-110) from the
parameters in tests/golden/ref_segment.json, in float32 like the Go code."""
import numpy as np

f32 = np.float32


def _box_points(width, length, height, res):
    """createBoxPoints (regiongrowing_test.go:17-27): float32 loop variables."""
    width, length, height, res = f32(width), f32(length), f32(height), f32(res)
    pts = []
    w = f32(-0.5) * width
    while w <= f32(0.5) * width:
        l = f32(-0.5) * length
        while l <= f32(0.5) * length:
            h = f32(-0.5) * height
            while h <= f32(0.5) * height:
                pts.append((w, l, h))
                h = f32(h + res)
            l = f32(l + res)
        w = f32(w + res)
    return np.array(pts, f32).reshape(-1, 3)


def region_growing_scene(g, seed=0):
    """-> (points float32 [n,3], labels uint32 [n], {object name: ids})."""
    rng = np.random.default_rng(seed)
    noise = f32(g["noise"])
    pts, labels, ids, cnt = [], [], {}, 0
    for o in g["objects"]:
        p = _box_points(*o["box"], o["res"])
        # min + rand.Float32()*(max-min)  (regiongrowing_test.go:96-98)
        nz = (-noise) + rng.random(p.shape, dtype=f32) * (noise - (-noise))
        p = (p + np.array(o["pos"], f32)) + nz
        pts.append(p.astype(f32))
        labels += [o["label"]] * len(p)
        ids[o["name"]] = list(range(cnt, cnt + len(p)))
        cnt += len(p)
    return np.ascontiguousarray(np.concatenate(pts), f32), np.array(labels, np.uint32), ids
