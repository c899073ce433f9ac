"""Generates tests/golden/visits.json: V(q), the mean number of tree nodes the
REFERENCE walk (kdtree.go:83-146,199-222) touches per query, counted by the
CPU oracle on the BASELINE configurations C2 (kNN 1M x 1M) and C4 (ICP 1M x 1M,
per iteration).  bench.py multiplies these by SURVEY.md 8(d)'s per-unit bytes
to obtain the algorithmic traffic of the roofline figure.

Run from the repo root:  python tests/golden/make_visits.py   (about 3 min, 1 core)
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle as O  # noqa: E402
from pcgol_amd import synth  # noqa: E402


def main():
    out = {"_about": "mean nodes touched / distance evaluations per query by the reference walk, "
                     "counted by oracle/pcgol_oracle.c (stat_visits / stat_dists); generator: "
                     "tests/golden/make_visits.py"}
    t0 = time.time()
    c2 = synth.c2_knn()
    tree = O.KDTree(c2["base"])
    ids, dsq, v, d = tree.nearest_batch(c2["queries"], c2["max_range"], stats=True)
    n = len(ids)
    out["c2_knn"] = {"n_base": len(c2["base"]), "n_query": n, "max_range": c2["max_range"],
                     "visits_per_query": v / n, "dists_per_query": d / n,
                     "ids_sum": int(ids.sum()), "ids_xor": int(np.bitwise_xor.reduce(ids)),
                     "dsq_bits_xor": int(np.bitwise_xor.reduce(dsq.view(np.uint32)))}
    print("c2", out["c2_knn"], time.time() - t0, flush=True)

    c4 = synth.c4_icp()
    trans = O.translate(0, 0, 0)
    it = 0
    per_iter = []
    tt = c4["target"].copy()
    for k in range(c4["max_iteration"]):
        _, _, v, d = tree.nearest_batch(tt, c4["max_dist"], stats=True)
        ev = O.icp_evaluate(tree, tt, c4["max_dist"], c4["min_pairs"])
        per_iter.append({"iteration": k, "visits_per_point": v / len(tt), "dists_per_point": d / len(tt),
                         "num_pairs": ev["npairs"], "value": float(ev["value"])})
        print(per_iter[-1], time.time() - t0, flush=True)
        trans, conv, it = O.icp_update(trans, ev["gradient"], it, c4["weight"], c4["threshold"], c4["max_iteration"])
        if conv:
            break
        tt = synth.transform_points(trans, c4["target"])
    out["c4_icp"] = {"n_base": len(c4["base"]), "n_target": len(c4["target"]), "max_dist": c4["max_dist"],
                     "per_iteration": per_iter,
                     "mean_visits_per_point": float(np.mean([p["visits_per_point"] for p in per_iter])),
                     "final_trans": [float(x) for x in trans]}
    with open(os.path.join(ROOT, "tests", "golden", "visits.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("done", time.time() - t0)


if __name__ == "__main__":
    main()
