#!/usr/bin/env python3
"""bench.py -- headline benchmark of the pcgol_amd hot path on MI355X.

Metric (BASELINE.json): Mpoints/s of one ICP iteration (correspondence +
reduction [+ re-projection + pose update]) on the 1M-point cloud, with the
kNN queries/s (C2) and VoxelGrid Mpoints/s (C3) figures measured alongside.

A "step" is ONE ICP iteration over this rank's tile of the target (1M points
per GPU: weak scaling) against the replicated 1M-point base KD-tree:
    partials kernel (transform + nearest + 10 partial sums)
    -> [N > 1] RCCL all-reduce of the 10 float64 sums (torch.distributed)
    -> update kernel (evaluate tail + gradient-descent pose update, on device)
Every 20 steps a new Fit starts (state reset), exactly as the reference's
MaxIteration = 20 loop (icp.go:48-65); Threshold = -1 keeps all iterations.

Prints ONE JSON line on rank 0 (see the driver contract in the task statement).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s achievable


def make_tile(synth, base, rank, world, n_per_gpu):
    """Rank's spatial tile of the global target (world x n_per_gpu points).
    Block b of the global target = T * base[perm_b] (perm seed 5 + b: block 0 is exactly
    config C4); the global cloud is Morton-sorted and cut into `world` contiguous ranges."""
    pose = synth.icp_pose()
    if world == 1:
        perm = np.random.Generator(np.random.PCG64(5)).permutation(len(base))[:n_per_gpu]
        return synth.transform_points(pose, base[perm])
    blocks = []
    for b in range(world):
        perm = np.random.Generator(np.random.PCG64(5 + b)).permutation(len(base))[:n_per_gpu]
        blocks.append(synth.transform_points(pose, base[perm]))
    g = np.concatenate(blocks)
    from pcgol_amd.distributed import spatial_tiles
    return np.ascontiguousarray(g[spatial_tiles(g, world)[rank]])


def load_visits():
    p = os.path.join(ROOT, "tests", "golden", "visits.json")
    with open(p) as f:
        return json.load(f)


def load_traffic(kernel="icp_corr_kernel"):
    """HBM bytes per launch of `kernel` from the newest committed PMC summary (profiles/rNN*_pmc.json,
    written by profiles/collect.sh: FETCH_SIZE and WRITE_SIZE collected in separate --pmc passes of this
    same bench command).  rocprofv3 reports both in KiB; on gfx950 FETCH_SIZE tallies 128-byte
    requests at 64 bytes, so it is doubled (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is exact.
    PMC collection cannot run inside the timed process, hence the committed summary."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9]*_pmc.json")))
    for p in reversed(files):
        try:
            with open(p) as f:
                d = json.load(f)
        except (OSError, ValueError):
            continue
        for name, c in d.items():
            if kernel in name and "FETCH_SIZE" in c and "WRITE_SIZE" in c:
                fetch = c["FETCH_SIZE"]["mean_per_dispatch"] * 1024.0 * 2.0
                write = c["WRITE_SIZE"]["mean_per_dispatch"] * 1024.0
                return fetch + write, os.path.basename(p)
    return None, None


def cpu_baseline(synth, base, target, cfg, budget_s=25.0):
    """The CPU oracle (C restatement of the reference algorithm, 1 thread) timed on the same
    workload: whole ICP iterations (corr + reduce + re-projection + update)."""
    import oracle as O
    tree = O.KDTree(base)  # build not timed (the GPU figure excludes it too)
    trans = O.translate(0, 0, 0)
    it = 0
    tt = target.copy()
    iters = 0
    t0 = time.perf_counter()
    while iters < cfg["max_iteration"]:
        ev = O.icp_evaluate(tree, tt, cfg["max_dist"], cfg["min_pairs"])
        trans, conv, it = O.icp_update(trans, ev["gradient"], it, cfg["weight"], cfg["threshold"],
                                       cfg["max_iteration"])
        tt = synth.transform_points(trans, target)
        iters += 1
        if conv or time.perf_counter() - t0 > budget_s * 0.7:
            break
    dt = time.perf_counter() - t0
    return {"value": len(target) * iters / dt / 1e6, "unit": "Mpoints/s", "cores": 1, "kind": "port",
            "sample": "%d full ICP iterations (corr+reduce+re-projection+update) of the 1M x 1M C4 workload, "
                      "oracle/pcgol_oracle.c, 1 thread, tree build excluded" % iters,
            "seconds": dt}


def side_benchmarks(torch, L, kdtree, voxelgrid, synth, tree, stream, visits, icp):
    """kNN (C2) and VoxelGrid (C3) throughput with inputs resident in HBM; reported as extras."""
    out = {}
    dev = "cuda"
    c2q = synth.uniform_cloud(1_000_000, 10.0, 3)
    dq = torch.from_numpy(c2q).to(dev)
    ids = torch.empty(len(c2q), dtype=torch.int32, device=dev)
    dsq = torch.empty(len(c2q), dtype=torch.float32, device=dev)
    for presort, key in ((True, "knn_c2_presort"), (False, "knn_c2_unsorted")):
        for _ in range(2):
            tree.NearestBatchDev(dq.data_ptr(), len(c2q), 10.0, ids.data_ptr(), dsq.data_ptr(), presort, stream)
        torch.cuda.synchronize()
        L.prof_reset()
        reps = 10
        t0 = time.perf_counter()
        for _ in range(reps):
            tree.NearestBatchDev(dq.data_ptr(), len(c2q), 10.0, ids.data_ptr(), dsq.data_ptr(), presort, stream)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        kms, kn = L.prof_read(L.PROF_KNN_WALK)
        gms, gn = L.prof_read(L.PROF_KNN_GRID)
        v = visits["c2_knn"]["visits_per_query"]
        ref = (12 + 8 + 16 * v) * len(c2q)  # bytes the reference's walk touches
        out[key] = {"mqueries_per_s": len(c2q) / dt / 1e6, "ms_per_call": dt * 1e3,
                    "walk_kernel_ms": kms / max(kn, 1)}
        if gn > 0:
            st = (C.c_int64 * 14)()
            L.check(L.lib().pcgx_debug_grid_stats(tree._h, L.ptr(dq.data_ptr()), len(c2q), 10.0, st))
            alg = (12 + 8 + 16 * st[12] / len(c2q) + 4 * st[13] / len(c2q)) * len(c2q)
            out[key].update({"grid_kernel_ms": gms / gn, "queries_left_to_walk": st[0],
                             "point_records_per_query": st[12] / len(c2q), "bound_words_per_query": st[13] / len(c2q),
                             "roofline_frac_grid_kernel": alg / (gms / gn * 1e-3) / 1e9 / HBM_PEAK_GBS,
                             "reference_walk_frac_of_peak": ref / ((gms / gn + kms / max(kn, 1)) * 1e-3) / 1e9 / HBM_PEAK_GBS})
        else:
            out[key]["roofline_frac_walk_kernel"] = ref / (kms / max(kn, 1) * 1e-3) / 1e9 / HBM_PEAK_GBS
    # Point-to-plane / Gauss-Newton extension (BASELINE.json config "ICP point-to-plane, 1M source vs
    # 1M target, 20 iters"; the reference has no such evaluator: no reference parity).
    # One step = correspondence + 30-sum reduction (6x6 normal equations) + Gauss-Newton update.
    from pcgol_amd.distributed import ShardedIcp
    cp = synth.c4_plane(1_000_000)
    ptree = kdtree.New(cp["base"])
    picp = ShardedIcp(ptree, cp["target"], cp["max_dist"], cp["min_pairs"], None, cp["threshold"],
                      cp["max_iteration"], BaseNormals=cp["normals"])
    for _ in range(20):
        picp.step()
    torch.cuda.synchronize()
    L.prof_reset()
    reps = 3
    t0 = time.perf_counter()
    for _ in range(reps):
        picp.reset()
        for _ in range(cp["max_iteration"]):
            picp.step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / (reps * cp["max_iteration"])
    kms, kn = L.prof_read(L.PROF_ICP_WALK)
    ptrans, pstat, _ = picp.result()
    inv = np.linalg.inv(synth.icp_pose().astype(np.float64).reshape(4, 4).T).T.reshape(-1)
    out["icp_plane_c4"] = {"mpoints_per_s": len(cp["target"]) / dt / 1e6, "ms_per_step": dt * 1e3,
                           "corr_kernel_ms": kms / max(kn, 1), "final_value": float(pstat.Evaluated.Value),
                           "pose_error_max": float(np.max(np.abs(ptrans.astype(np.float64) - inv))),
                           "exchange_doubles": 30, "parity": "none in the reference (extension)"}
    picp.close()
    del ptree
    # STRICT sums (sequential float32 in target order: bit-identical to the Go code at any size)
    c4 = synth.c4_icp()
    ss = icp.IcpSession(tree, c4["target"], c4["max_dist"], c4["min_pairs"], c4["weight"], c4["threshold"],
                        c4["max_iteration"])
    ss.set_strict(True)
    for _ in range(3):
        ss.step(stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        ss.step(stream)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    out["icp_strict_c4"] = {"mpoints_per_s": len(c4["target"]) / dt / 1e6, "ms_per_step": dt * 1e3,
                            "note": "one wave adds the evaluator's float32 terms sequentially (reference bits)"}
    ss.close()
    c3 = synth.c3_voxel()
    dp = torch.from_numpy(c3["points"]).to(dev)
    dout = torch.empty_like(dp)
    vg = voxelgrid.New(c3["leaf"])
    for _ in range(2):
        m = vg.FilterDev(dp.data_ptr(), len(c3["points"]), 12, 0, dout.data_ptr(), stream)
    torch.cuda.synchronize()
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps):
        m = vg.FilterDev(dp.data_ptr(), len(c3["points"]), 12, 0, dout.data_ptr(), stream)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    n = len(c3["points"])
    alg = 24 * n + 24 * m
    out["voxel_c3"] = {"mpoints_per_s": n / dt / 1e6, "ms_per_call": dt * 1e3, "out_points": int(m),
                       "algorithmic_gbs": alg / dt / 1e9, "roofline_frac": alg / dt / 1e9 / HBM_PEAK_GBS}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--points", type=int, default=1_000_000, help="target points per GPU (and base size)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d"
                     % (args.gpus, args.gpus))
        args.gpus = world

    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (no CPU fallback)")
    # Rehearsal of the N > 1 path on a one-GPU box: PCGX_BENCH_REHEARSE=1 puts every rank on
    # cuda:0 and exchanges through gloo (RCCL refuses two ranks on one device).  Never used by the
    # driver; the numbers of such a run are not comparable (ranks share the GPU).
    rehearse = os.environ.get("PCGX_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from pcgol_amd import _lib as L
    from pcgol_amd import icp, kdtree, synth, voxelgrid
    L.check(L.lib().pcgx_init(local_rank))

    n = args.points
    width = 10.0 * (n / 1e6) ** (1.0 / 3.0)  # keeps the C4 point density when --points is changed
    base = synth.uniform_cloud(n, width, 2)
    cfg = dict(max_dist=0.5, min_pairs=6, weight=np.full(6, 0.3, np.float32),
               threshold=np.full(6, -1.0, np.float32), max_iteration=20)
    tile = make_tile(synth, base, rank, world, n)
    t0 = time.perf_counter()
    tree = kdtree.New(base)
    build_s = time.perf_counter() - t0

    side_stream = torch.cuda.Stream()  # a real (non-zero) stream handle for the side benchmarks
    stream = side_stream.cuda_stream
    from pcgol_amd.distributed import ShardedIcp
    sicp = ShardedIcp(tree, tile, cfg["max_dist"], cfg["min_pairs"], cfg["weight"], cfg["threshold"],
                      cfg["max_iteration"])
    in_fit = [0]

    def step():
        if in_fit[0] == cfg["max_iteration"]:
            sicp.reset()
            in_fit[0] = 0
        sicp.step()  # partials kernel -> [N>1: RCCL all-reduce of 80 bytes] -> update kernel
        in_fit[0] += 1

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    # HIP events around every 3rd launch of the timed kernels (3 is coprime to the 20 iterations of a
    # Fit, so every iteration index is sampled): a pair of events costs ~5 us of stream time next to a
    # 30 us kernel, timing each launch would lower `value` by a fifth
    L.prof_enable(3)
    L.prof_reset()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    barrier()
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    walk_ms, walk_n = L.prof_read(L.PROF_ICP_WALK)
    grid_ms, grid_n = L.prof_read(L.PROF_ICP_GRID)
    trans, stat, _ = sicp.result()

    # What the grid pass reads per iteration (an untimed Fit, one instrumented launch before each
    # step): point records and cell-bound words, targets left to the tree walk.
    grid_pts = grid_words = grid_walked = 0
    if grid_n > 0:
        sicp.reset()
        for _ in range(cfg["max_iteration"]):
            g = sicp.sess.grid_stats(sicp.stream.cuda_stream)
            grid_walked += g[1]
            grid_pts += g[2]
            grid_words += g[3]
            sicp.step()
        torch.cuda.synchronize()

    if rank == 0:
        visits = load_visits()
        v_icp = visits["c4_icp"]["mean_visits_per_point"]
        ref_bytes = (12 + 16 * v_icp) * n  # 12 B target read + 16 B per node the REFERENCE walk touches
        if grid_n > 0:
            # the dominant kernel is the grid pass; its algorithmic bytes in SURVEY 8(d)'s form, for
            # the records IT touches: 12 B target + 16 B previous pair read (iterations >= 1) + 16 B
            # pair written + 16 B per point record + 4 B per cell-bound word, averaged over a Fit
            its = cfg["max_iteration"]
            v_pts, v_words = grid_pts / (its * n), grid_words / (its * n)
            alg_bytes = (12 + 16 * (its - 1) / its + 16 + 16 * v_pts + 4 * v_words) * n
            kernel, kernel_s, launches = "icp_grid_kernel", grid_ms / max(grid_n, 1) * 1e-3, grid_n
        else:
            v_pts = v_words = None
            alg_bytes = ref_bytes
            kernel, kernel_s, launches = "icp_corr_kernel", walk_ms / max(walk_n, 1) * 1e-3, walk_n
        achieved = alg_bytes / kernel_s / 1e9
        traffic, traffic_src = load_traffic(kernel) if n == 1_000_000 else (None, None)
        corr_s = walk_ms / max(walk_n, 1) * 1e-3 if grid_n == 0 else None  # with the grid pass: see profiles/
        line = {
            "metric": "Mpoints/sec ICP iter (corr+reduce) + kNN queries/sec, 1M-pt cloud",
            "value": world * n * args.steps / elapsed / 1e6,
            "unit": "Mpoints/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "C4 ICP point-to-point gradient iteration (reference has no point-to-plane): "
                                   "%d-pt base KD-tree (replicated) x %d target pts per GPU, MaxDist 0.5, "
                                   "20-iteration Fits, Threshold -1; one step = corr+reduce+re-projection+update"
                                   % (n, n),
                       "base_points": n, "target_points_per_gpu": n, "parallelism": "target tiles x%d, tree replicated" % world,
                       "exchange": "none" if world == 1 else "all-reduce 10 x f64 per step (%s)"
                                   % ("gloo: REHEARSAL on one GPU, not a measurement" if rehearse else "RCCL")},
            "roofline": {"bound": "hbm", "kernel": kernel, "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_src, "kernel_ms": kernel_s * 1e3, "launches": launches,
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "point_records_per_target": v_pts, "bound_words_per_target": v_words,
                         "targets_left_to_walk_per_fit": grid_walked if grid_n > 0 else None,
                         "corr_kernel_ms": corr_s * 1e3 if corr_s is not None else None,
                         # the same launch priced as SURVEY 8(d) prices the reference's walk (12 + 16 V B per
                         # target, V = nodes the reference touches): above 1 means the grid answers with
                         # fewer bytes than that walk would stream
                         "reference_walk": {"visits_per_point": v_icp, "bytes_per_launch": ref_bytes,
                                            "rate_gbs": ref_bytes / kernel_s / 1e9,
                                            "frac_of_peak": ref_bytes / kernel_s / 1e9 / HBM_PEAK_GBS}},
            "tree_build_s": build_s,
            "final_value": float(stat.Evaluated.Value),
        }
        if world == 1 and not args.no_extras:
            line["extra"] = side_benchmarks(torch, L, kdtree, voxelgrid, synth, tree, stream, visits, icp)
            # the second half of BASELINE.json's metric ("+ kNN queries/sec, 1M-pt cloud", config C2)
            line["knn_queries_per_s"] = line["extra"]["knn_c2_presort"]["mqueries_per_s"] * 1e6
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(synth, base, tile, cfg)
            line["cpu_baseline"]["host_cpus"] = os.cpu_count()
        print(json.dumps(line), flush=True)
    sicp.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
