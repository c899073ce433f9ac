"""Synthetic inputs of the BASELINE.json configurations.

Coordinates mirror the reference's own random clouds, `rand.Float32()*width`
(pc/storage/kdtree/kdtree_test.go:1007-1013): u = k / 2^24 with k uniform in
[0, 2^24), x = float32(u * W).  Generator: numpy PCG64 with the stated seed."""
import numpy as np


def uniform_cloud(n, width, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    k = rng.integers(0, 1 << 24, size=(n, 3), dtype=np.int64)
    u = k.astype(np.float32) / np.float32(1 << 24)
    return np.ascontiguousarray(u * np.float32(width), dtype=np.float32)


def c1_voxel():   # VoxelGrid 100k, leaf 0.05 (CPU reference config)
    return dict(points=uniform_cloud(100_000, 1.6, 1), leaf=(0.05, 0.05, 0.05))


def c2_knn(n_base=1_000_000, n_query=1_000_000):   # kNN k=1, 1M x 1M, width 10, maxRange 10
    return dict(base=uniform_cloud(n_base, 10.0, 2), queries=uniform_cloud(n_query, 10.0, 3), max_range=10.0)


def c3_voxel(n=10_000_000):   # VoxelGrid 10M, leaf 0.02, cube 3.0 m
    return dict(points=uniform_cloud(n, 3.0, 4), leaf=(0.02, 0.02, 0.02))


def icp_pose():
    """T = Translate(.02,.01,-.015) * Rotate(0,0,1,.001) in the reference's float32 arithmetic
    (mat/transform.go:7-14,25-35; mat/mat4.go:16-28), column-major."""
    f = np.float32
    ang = f(0.001)
    s, c = f(np.sin(np.float64(ang))), f(np.cos(np.float64(ang)))
    one_c = f(f(1) - c)
    x, y, z = f(0), f(0), f(1)
    r = np.array([
        c + x * x * one_c, x * y * one_c + z * s, x * z * one_c - y * s, 0,
        y * x * one_c - z * s, c + y * y * one_c, y * z * one_c + x * s, 0,
        z * x * one_c + y * s, z * y * one_c - x * s, c + z * z * one_c, 0,
        0, 0, 0, 1], dtype=np.float32)
    t = np.eye(4, dtype=np.float32).reshape(-1)
    t[12:15] = (f(0.02), f(0.01), f(-0.015))
    # Mat4.Mul (t * r): out[4j+i] = sum_k t[4k+i] * r[4j+k], float32 left to right
    out = np.zeros(16, np.float32)
    for i in range(4):
        for j in range(4):
            acc = f(0)
            for k in range(4):
                acc = f(acc + f(t[4 * k + i] * r[4 * j + k]))
            out[4 * j + i] = acc
    return out


def transform_points(m, pts):
    """Mat4.Transform (mat/mat4.go:130-137) vectorised in float32, left to right."""
    f = np.float32
    x, y, z = pts[:, 0], pts[:, 1], pts[:, 2]
    w = f(1) / (((m[3] * x + m[7] * y) + m[11] * z) + m[15])
    ox = (((m[0] * x + m[4] * y) + m[8] * z) + m[12]) * w
    oy = (((m[1] * x + m[5] * y) + m[9] * z) + m[13]) * w
    oz = (((m[2] * x + m[6] * y) + m[10] * z) + m[14]) * w
    return np.ascontiguousarray(np.stack([ox, oy, oz], axis=1), dtype=np.float32)


def c4_icp(n=1_000_000, width=10.0, base_seed=2, perm_seed=5):
    """ICP 1M x 1M: target_i = T * base[perm(i)]; MaxDist 0.5, MinPairs 6, Weight 0.3,
    Threshold -1 (run all iterations, as icp_test.go:126), MaxIteration 20."""
    base = uniform_cloud(n, width, base_seed)
    perm = np.random.Generator(np.random.PCG64(perm_seed)).permutation(n)
    target = transform_points(icp_pose(), base[perm])
    return dict(base=base, target=target, max_dist=0.5, min_pairs=6,
                weight=np.full(6, 0.3, np.float32), threshold=np.full(6, -1.0, np.float32),
                max_iteration=20)


def surface_cloud(n, width, seed):
    """Points on the smooth surface z = h(x, y) over [0, width)^2 with their analytic unit
    normals (input of the point-to-plane extension; the uniform cube has no surface).
    h = 0.5 sin(0.7 x) cos(0.5 y) + 0.3 sin(1.3 y): curved in both directions, so the
    point-to-plane normal equations constrain all six pose parameters."""
    rng = np.random.Generator(np.random.PCG64(seed))
    k = rng.integers(0, 1 << 24, size=(n, 2), dtype=np.int64)
    xy = (k.astype(np.float64) / float(1 << 24)) * float(width)
    x, y = xy[:, 0], xy[:, 1]
    z = 0.5 * np.sin(0.7 * x) * np.cos(0.5 * y) + 0.3 * np.sin(1.3 * y)
    hx = 0.35 * np.cos(0.7 * x) * np.cos(0.5 * y)
    hy = -0.25 * np.sin(0.7 * x) * np.sin(0.5 * y) + 0.39 * np.cos(1.3 * y)
    nrm = np.stack([-hx, -hy, np.ones_like(x)], axis=1)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    pts = np.stack([x, y, z], axis=1)
    return (np.ascontiguousarray(pts, dtype=np.float32), np.ascontiguousarray(nrm, dtype=np.float32))


def c4_plane(n=1_000_000, width=None, base_seed=6, perm_seed=7):
    """Point-to-plane variant of C4 (BASELINE.json "ICP point-to-plane, 1M source vs 1M target,
    20 iters"): base = n surface points with normals, target_i = T * base[perm(i)], T = icp_pose();
    MaxDist 0.5, MinPairs 6, Threshold -1 (all iterations), MaxIteration 20."""
    if width is None:
        width = 30.0 * (n / 1e6) ** 0.5  # ~1100 points per square metre at any n
    base, normals = surface_cloud(n, width, base_seed)
    perm = np.random.Generator(np.random.PCG64(perm_seed)).permutation(n)
    target = transform_points(icp_pose(), base[perm])
    return dict(base=base, normals=normals, target=target, max_dist=0.5, min_pairs=6,
                threshold=np.full(6, -1.0, np.float32), max_iteration=20, damping=0.0)


def spatial_cell(points, world, lo, hi):
    """Which of `world` spatial tiles of the box [lo, hi) each point falls in -- decided point by
    point, nothing sorted, so a rank finds its own tile without looking at anybody else's.  The box is
    cut along x, then y, then z, then x again ... by successive halving while the factor 2 divides
    `world` (8 tiles = the octants), and into equal slabs along the next axis for what is left (an
    odd factor).  Uniform clouds give equally filled tiles up to sampling noise."""
    points = np.asarray(points, np.float32).reshape(-1, 3)
    cell = np.zeros(len(points), np.int64)
    stride = 1
    rest = int(world)
    axis = 0
    u = (points.astype(np.float64) - np.asarray(lo, np.float64)) / (np.asarray(hi, np.float64) - np.asarray(lo, np.float64))
    while rest > 1:
        k = 2 if rest % 2 == 0 else rest
        c = np.clip((u[:, axis % 3] * k).astype(np.int64), 0, k - 1)
        if axis >= 3:   # this axis was cut before: cut each of its parts again
            prev = 2 ** (axis // 3)
            c = np.clip((u[:, axis % 3] * prev * k).astype(np.int64), 0, prev * k - 1) % k
        cell += stride * c
        stride *= k
        rest //= k
        axis += 1
    return cell


def icp_tile(base, rank, world, n_per_gpu, width, perm_seed=5, order_seed=17):
    """Rank's spatial tile of the global target of a `world`-GPU ICP job: the global target is
    `world` blocks T * base[perm_b][:n_per_gpu] (perm seed perm_seed + b; block 0 alone is config C4),
    i.e. world * n_per_gpu points in the base's box; the rank keeps the points of ITS cell
    (spatial_cell) in a seeded random order.  world == 1: exactly c4_icp's target."""
    pose = icp_pose()
    if world == 1:
        perm = np.random.Generator(np.random.PCG64(perm_seed)).permutation(len(base))[:n_per_gpu]
        return transform_points(pose, base[perm])
    parts = []
    for b in range(world):
        perm = np.random.Generator(np.random.PCG64(perm_seed + b)).permutation(len(base))[:n_per_gpu]
        t = transform_points(pose, base[perm])
        parts.append(t[spatial_cell(t, world, (0, 0, 0), (width,) * 3) == rank])
    t = np.concatenate(parts)
    order = np.random.Generator(np.random.PCG64(order_seed + rank)).permutation(len(t))
    return np.ascontiguousarray(t[order])


def c5_tile(base, rank, world=8, width=40.0, order_seed=23, chunk=8_000_000):
    """Config C5 (ICP on a 64M-point cloud tiled over 8 GPUs), the share of one rank: the global
    target is T * base (every base point once); the rank keeps the points of its spatial cell
    (world = 8: an octant of the cube), in a seeded random order -- a pass over the base in chunks,
    no global permutation or sort.  ~len(base) / world points."""
    pose = icp_pose()
    parts = []
    for s in range(0, len(base), chunk):
        t = transform_points(pose, base[s:s + chunk])
        parts.append(t[spatial_cell(t, world, (0, 0, 0), (width,) * 3) == rank])
    t = np.concatenate(parts)
    order = np.random.Generator(np.random.PCG64(order_seed + rank)).permutation(len(t))
    return np.ascontiguousarray(t[order])


def uniform_cloud_chunked(n, width, seed, chunk=8_000_000):
    """uniform_cloud for sizes where its int64 temporaries would be large (same distribution; the
    stream differs from uniform_cloud's because numbers are drawn chunk by chunk)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    out = np.empty((n, 3), np.float32)
    for s in range(0, n, chunk):
        m = min(chunk, n - s)
        k = rng.integers(0, 1 << 24, size=(m, 3), dtype=np.int32)
        out[s:s + m] = (k.astype(np.float32) / np.float32(1 << 24)) * np.float32(width)
    return out
