"""C5-shaped single-rank run: 64M-point base tree (what every rank of the 8-GPU job holds) and one
8M-point target tile.  Checks a sample of nearest-neighbour answers against brute force (torch,
same float32 expression order) and times tree build, a kNN batch and ICP iterations.
    python tools/c5_probe.py [n_base] [n_tile]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from pcgol_amd import _lib as L
from pcgol_amd import kdtree, synth
from pcgol_amd.distributed import ShardedIcp

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 64_000_000
nt = int(sys.argv[2]) if len(sys.argv) > 2 else 8_000_000
width = 10.0 * (nb / 1e6) ** (1.0 / 3.0)
L.check(L.lib().pcgx_init(0))
t0 = time.perf_counter()
base = synth.uniform_cloud(nb, width, 2)
print("synth %.1f s, width %.2f" % (time.perf_counter() - t0, width), flush=True)
t0 = time.perf_counter()
tree = kdtree.New(base)
print("tree build (upload + device build + directory) %.3f s, depth %d" % (time.perf_counter() - t0, tree.MaxDepth()),
      flush=True)
perm = np.random.Generator(np.random.PCG64(5)).permutation(nb)[:nt]
tile = synth.transform_points(synth.icp_pose(), base[perm])

# --- brute-force check of a sample
ns = 256
q = tile[:: nt // ns][:ns].copy()
ids, dsq = tree.NearestBatch(q, 0.5)
dq = torch.from_numpy(q).cuda()
best_d = torch.full((ns,), float("inf"), device="cuda")
best_i = torch.full((ns,), -1, dtype=torch.int64, device="cuda")
chunk = 2_000_000
for s in range(0, nb, chunk):
    b = torch.from_numpy(base[s:s + chunk]).cuda()
    d = b[None, :, :] - dq[:, None, :]
    d2 = d * d
    dist = (d2[..., 0] + d2[..., 1]) + d2[..., 2]
    m, i = dist.min(dim=1)
    upd = m < best_d
    best_d = torch.where(upd, m, best_d)
    best_i = torch.where(upd, i + s, best_i)
bd, bi = best_d.cpu().numpy(), best_i.cpu().numpy()
inr = bd <= np.float32(0.5) * np.float32(0.5)
ok_d = np.array_equal(dsq[inr], bd[inr])
ok_i = np.array_equal(ids[inr], bi[inr])
print("brute-force sample: %d queries, %d in range, DistSq bit-equal %s, ids equal %s, out-of-range -> -1: %s" %
      (ns, int(inr.sum()), ok_d, ok_i, bool(np.all(ids[~inr] == -1))), flush=True)
assert ok_d and bool(np.all(ids[~inr] == -1))

# --- ICP iterations on the tile
cfg = dict(max_dist=0.5, min_pairs=6, weight=np.full(6, 0.3, np.float32), threshold=np.full(6, -1.0, np.float32),
           max_iteration=20)
sicp = ShardedIcp(tree, tile, cfg["max_dist"], cfg["min_pairs"], cfg["weight"], cfg["threshold"], cfg["max_iteration"])
L.prof_enable(True)
for rep in range(2):
    sicp.reset()
    torch.cuda.synchronize()
    L.prof_reset()
    t0 = time.perf_counter()
    for _ in range(20):
        sicp.step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ms, k = L.prof_read(L.PROF_ICP_GRID)
    which = "grid kernel"
    if k == 0:  # base tree without a grid: the walk kernel does the work
        ms, k = L.prof_read(L.PROF_ICP_WALK)
        which = "corr kernel"
    print("fit %d: 20 iterations %.2f ms (%.1f Mpoints/s), %s %.3f ms" %
          (rep, dt * 1e3, nt * 20 / dt / 1e6, which, ms / max(k, 1)), flush=True)
trans, stat, _ = sicp.result()
err = np.abs(np.asarray(trans, np.float64).reshape(4, 4) - np.linalg.inv(synth.icp_pose().astype(np.float64).reshape(4, 4).T).T)
print("final Value %.3e, |trans - inverse pose|max %.2e" % (float(stat.Evaluated.Value), float(err.max())))
sicp.close()
