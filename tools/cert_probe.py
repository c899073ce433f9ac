"""How many of C4's targets could keep last iteration's partner on a certificate alone: the partner p_i is the nearest
base point of target q as soon as |q - p_i| < r_i, r_i = half the distance from p_i to ITS nearest neighbour (any other
point is then farther from q than p_i is).  Per iteration of a C4 Fit: the share of a 50k sample with |q - p_i| <
0.99 r_i, with q moved by the pose the device's Fit has after that iteration.   python tools/cert_probe.py"""
import sys
import numpy as np
sys.path.insert(0, ".")
from scipy.spatial import cKDTree
from pcgol_amd import icp, kdtree, synth, _lib as L
c = synth.c4_icp()
base, target = c["base"], c["target"]
perm = np.random.Generator(np.random.PCG64(5)).permutation(len(base))
tree = cKDTree(base)
rng = np.random.default_rng(1)
sample = rng.choice(len(target), 50_000, replace=False)
partner = perm[sample]
d2, _ = tree.query(base[partner], k=2)
r = 0.5 * d2[:, 1]
print("half nearest-neighbour distance of the partners: mean %.4f median %.4f" % (r.mean(), np.median(r)))
t = kdtree.New(base)
s = icp.IcpSession(t, target, c["max_dist"], c["min_pairs"], c["weight"], c["threshold"], c["max_iteration"])
for it in range(20):
    trans, st, _ = s.result()
    T = np.asarray(trans, np.float64).reshape(4, 4).T
    q = target[sample].astype(np.float64) @ T[:3, :3].T + T[:3, 3]
    d = np.linalg.norm(q - base[partner], axis=1)
    dn, idx = tree.query(q, k=1)
    print("iteration %2d: offset to the true partner mean %.4f; nearest IS the partner %.3f; certified by 0.99 r: %.3f, by 0.5 r: %.3f" %
          (it, d.mean(), float(np.mean(idx == partner)), float(np.mean(d < 0.99 * r)), float(np.mean(d < 0.5 * r))))
    s.step()
