"""Generates tests/golden/c5_digest.json: what the CPU oracle (the C restatement of the reference,
oracle/pcgol_oracle.c) computes on BASELINE config C5 -- the 64M-point base cloud of the 8-GPU ICP job
and octant 0 of its target (synth.c5_tile) -- condensed into digests the GPU suite compares against
(tests/test_gpu_c5.py), because building the oracle's tree at this size takes minutes:

  nearest   the first 100 000 targets of the tile against the 64M-point tree (kdtree.go:83-146):
            xor / sum of the ids, xor of the DistSq bit patterns, number found, the first 64 ids
  pairs     iteration 0's correspondences of the first 1 000 000 targets (correspondence.go:22-37):
            pair count, xor of base ids
  sums      iteration 0's evaluator sums over those pairs (evaluator.go:122-145): the ten float64 sums
            of the float32 terms (sums_mode 1: what a sharded rank computes and all-reduces) and the
            reference's sequential float32 sums (sums_mode 0), Value / Gradient / DistRMS of Evaluate

  tile      the WHOLE ~8M-target tile (3907 tiles of the strict sums, eight chunks of the chain kernel): iteration 0's
            Evaluate with the reference's sequential float32 sums, and a three-iteration Fit (MaxIteration 3,
            Threshold -1): pose, Value, Gradient, DistRMS bits -- what test_c5_strict_sums_on_the_tile pins the
            device's parallel sums to (before round 5 that test compared them with the device's own one-wave chain)

Run from the repo root:  python tests/golden/make_c5_digest.py   (about 20 min, 1 core, 5 GB)
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

NB, WIDTH = 64_000_000, 40.0


def main():
    t0 = time.time()
    base = synth.uniform_cloud_chunked(NB, WIDTH, 2)
    tile = synth.c5_tile(base, 0, 8, WIDTH)
    print("clouds", len(base), len(tile), time.time() - t0, flush=True)
    tree = O.KDTree(base)
    print("oracle tree built", time.time() - t0, flush=True)
    out = {"_about": "digests of the CPU oracle's results on config C5 (64M-point base, octant 0 of the target); "
                     "generator: tests/golden/make_c5_digest.py",
           "n_base": NB, "width": WIDTH, "n_tile": len(tile), "max_dist": 0.5}
    q = np.ascontiguousarray(tile[:100_000])
    ids, dsq = tree.nearest_batch(q, 0.5)
    out["nearest"] = {"n": len(q), "found": int((ids >= 0).sum()), "ids_sum": int(ids.sum()),
                      "ids_xor": int(np.bitwise_xor.reduce(ids)),
                      "dsq_bits_xor": int(np.bitwise_xor.reduce(dsq.view(np.uint32))),
                      "first_ids": [int(v) for v in ids[:64]]}
    print("nearest", out["nearest"]["found"], time.time() - t0, flush=True)
    t1m = np.ascontiguousarray(tile[:1_000_000])
    b, t, d = O.icp_pairs(tree, t1m, 0.5)
    out["pairs"] = {"n_target": len(t1m), "n_pairs": len(b), "base_ids_xor": int(np.bitwise_xor.reduce(b)),
                    "target_ids_sum": int(t.sum()), "dsq_bits_xor": int(np.bitwise_xor.reduce(d.view(np.uint32)))}
    print("pairs", len(b), time.time() - t0, flush=True)
    e64 = O.icp_evaluate(tree, t1m, 0.5, 6, sums_mode=1)
    e32 = O.icp_evaluate(tree, t1m, 0.5, 6, sums_mode=0)
    out["sums"] = {"f64_tree_raw10": [float(v) for v in e64["raw10"]],
                   "reference_raw10": [float(v) for v in e32["raw10"]],
                   "reference_value_bits": int(np.float32(e32["value"]).view(np.uint32)),
                   "reference_gradient_bits": [int(v) for v in e32["gradient"].view(np.uint32)],
                   "reference_dist_rms_bits": int(np.float32(e32["dist_rms"]).view(np.uint32))}
    et = O.icp_evaluate(tree, tile, 0.5, 6, sums_mode=0)
    print("tile evaluate", et["npairs"], time.time() - t0, flush=True)
    w, th = np.full(6, 0.3, np.float32), np.full(6, -1.0, np.float32)
    ft = O.icp_fit(tree, tile, 0.5, 6, w, th, 3, sums_mode=0)
    print("tile fit", ft["num_iteration"], time.time() - t0, flush=True)
    out["tile"] = {"n_pairs": int(et["npairs"]),
                   "evaluate_value_bits": int(np.float32(et["value"]).view(np.uint32)),
                   "evaluate_gradient_bits": [int(v) for v in et["gradient"].view(np.uint32)],
                   "evaluate_dist_rms_bits": int(np.float32(et["dist_rms"]).view(np.uint32)),
                   "fit3_num_iteration": int(ft["num_iteration"]),
                   "fit3_trans_bits": [int(v) for v in np.asarray(ft["trans"], np.float32).ravel().view(np.uint32)],
                   "fit3_value_bits": int(np.float32(ft["value"]).view(np.uint32)),
                   "fit3_gradient_bits": [int(v) for v in np.asarray(ft["gradient"], np.float32).view(np.uint32)],
                   "fit3_dist_rms_bits": int(np.float32(ft["dist_rms"]).view(np.uint32))}
    with open(os.path.join(ROOT, "tests", "golden", "c5_digest.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("done", time.time() - t0)


if __name__ == "__main__":
    main()
