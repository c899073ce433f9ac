#!/usr/bin/env python3
"""bench.py -- headline benchmark of the pcgol_amd hot path on MI355X.

Metric (BASELINE.json): Mpoints/s of one ICP iteration (correspondence + reduction [+ re-projection
+ pose update]) on the 1M-point cloud, with the kNN queries/s (C2) and VoxelGrid Mpoints/s (C3)
figures measured alongside.

A "step" is ONE ICP iteration over this rank's tile of the target against the replicated base
KD-tree.  Every 20 steps a new Fit starts (state reset), as in the reference's MaxIteration = 20 loop
(icp.go:48-65); Threshold = -1 keeps all iterations.

  --workload c4 (default)  BASELINE's fourth config: 1M-point base, 1M target points per GPU (weak scaling)
  --workload c5            BASELINE's fifth config: the 64M-point base replicated on every GPU, rank r
                           holds octant r of the 64M-point target (synth.c5_tile); at --gpus 8 this is
                           the config as worded, at fewer GPUs the first N octants

  N = 1   the drop-in default, which MATCHES THE REFERENCE BIT FOR BIT ("parity_mode": "reference"):
          certified grid pass -> leftover walk (+ tile sums) -> the evaluator's sequential float32 sums
          evaluated exactly in parallel (csrc/strict_sum.h) -> pose update, all enqueued on the device.
          The same GPU's float64-tree figure is on the line as `value_f64_tree`.
  N > 1   one spatial tile of the target per GPU, 10 float64 partial sums, ONE all-reduce per step
          through the library's own RCCL communicator (pcgx_icp_session_step_sharded).  A sum
          spread over ranks has no sequential order: "parity_mode": "f64-tree" (float64 reduction of
          the reference's float32 terms; differs from the Go code by ITS rounding noise).
          `value_same_mode_n1` is what ONE of these GPUs does in the same mode without the exchange:
          the base of a scaling curve in one numeric mode.

The timed loop runs with profiling off; the kernel times of `roofline` come from a second, untimed pass.
Launched as `python bench.py --gpus N` this script starts the N ranks itself
(torch.distributed.run as a child process, before anything touches the GPU); launched under
torch.distributed.run it is one of those ranks.  Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); the copy microbench below gives the reachable rate


def load_visits():
    with open(os.path.join(ROOT, "tests", "golden", "visits.json")) as f:
        return json.load(f)


def load_pmc_summary():
    """The newest committed PMC summary (profiles/rNN*_pmc.json, written by profiles/collect.sh +
    summarize.py from separate --pmc passes of this same bench command) and whether it was collected on
    the build that is being timed (its `_build.source_hash` against pcgol_amd.build.source_hash())."""
    import glob
    from pcgol_amd import build
    for p in reversed(sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9]*_pmc.json")))):
        try:
            with open(p) as f:
                d = json.load(f)
        except (OSError, ValueError):
            continue
        meta = d.get("_build", {})
        return d, os.path.basename(p), meta.get("source_hash") == build.source_hash()
    return {}, None, False


def traffic_of(summary, kernel):
    """HBM bytes per launch of `kernel` from a PMC summary.  rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB;
    FETCH_SIZE is doubled for wide streaming reads only (gfx950 tallies their 128-byte requests at 64 bytes,
    MI355X_MICROARCH.md); a random gather is reported at the sector bytes it costs (profiles/rNN_fetch_probe.json),
    so gather kernels take the counter as it is plus half of their known coalesced reads (summarize.py);
    WRITE_SIZE is exact."""
    for name, c in sorted(summary.items(), key=lambda kv: -len(kv[0])):  # the most specific entry first
        if kernel in name and isinstance(c, dict) and "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            scale = float(c.get("fetch_scale", 2.0))
            fetch = c["FETCH_SIZE"]["mean_per_dispatch"] * 1024.0 * scale + float(c.get("fetch_add_bytes", 0.0))
            write = c["WRITE_SIZE"]["mean_per_dispatch"] * 1024.0
            return fetch + write
    return None


def load_traffic(kernel):
    summary, src, _ = load_pmc_summary()
    return traffic_of(summary, kernel), src


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(synth, base, target, cfg, budget_s=25.0):
    """The CPU oracle (C restatement of the reference algorithm, 1 thread: the Go reference has no
    goroutines on this path) timed on the same workload: whole ICP iterations (corr + reduce +
    re-projection + update)."""
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
            "seconds": dt, "cpu_model": cpu_model(), "host_cpus": os.cpu_count()}


def hbm_copy_gbs(torch):
    """Reachable HBM rate of this GPU: device-to-device copy of 1 GiB (read + write counted)."""
    n = 1 << 28
    a = torch.empty(n, dtype=torch.float32, device="cuda")
    b = torch.empty(n, dtype=torch.float32, device="cuda")
    a.fill_(1.0)
    for _ in range(3):
        b.copy_(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    e0.record()
    for _ in range(reps):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    del a, b
    return 2.0 * n * 4 / (ms * 1e-3) / 1e9


def time_session_steps(torch, L, sess, steps, max_iteration, stream=0):
    """steps ICP iterations of a device-resident session (a new Fit every max_iteration), wall time."""
    k = 0
    L.check(L.lib().pcgx_icp_session_reset(sess._h, L.ptr(stream) if stream else None))
    torch.cuda.synchronize()
    L.check(L.lib().pcgx_sync(None))
    t0 = time.perf_counter()
    for _ in range(steps):
        if k == max_iteration:
            L.check(L.lib().pcgx_icp_session_reset(sess._h, L.ptr(stream) if stream else None))
            k = 0
        sess.step(stream)
        k += 1
    L.check(L.lib().pcgx_sync(L.ptr(stream) if stream else None))
    torch.cuda.synchronize()
    return time.perf_counter() - t0


def side_benchmarks(torch, L, kdtree, voxelgrid, synth, tree, stream, visits, icp, hbm_gbs):
    """kNN (C2), VoxelGrid (C3), the f64-tree ICP mode and the point-to-plane extension, inputs
    resident in HBM; reported as extras."""
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
        reps = 20
        t0 = time.perf_counter()
        for _ in range(reps):
            tree.NearestBatchDev(dq.data_ptr(), len(c2q), 10.0, ids.data_ptr(), dsq.data_ptr(), presort, stream)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        # kernel times from a pass of their own: the events around every kernel cost a call ~30 us
        L.prof_enable(1)
        L.prof_reset()
        for _ in range(10):
            tree.NearestBatchDev(dq.data_ptr(), len(c2q), 10.0, ids.data_ptr(), dsq.data_ptr(), presort, stream)
        torch.cuda.synchronize()
        L.prof_enable(0)
        kms, kn = L.prof_read(L.PROF_KNN_WALK)
        gms, gn = L.prof_read(L.PROF_KNN_GRID)
        v = visits["c2_knn"]["visits_per_query"]
        ref = (12 + 8 + 16 * v) * len(c2q)  # bytes the reference's walk touches
        out[key] = {"mqueries_per_s": len(c2q) / dt / 1e6, "ms_per_call": dt * 1e3,
                    "walk_kernel_ms": kms / max(kn, 1)}
        if gn > 0:
            kernel_s = gms / gn * 1e-3
            if presort:   # the search kernel of the partitioned path; the whole call's bytes beside it
                traffic, src = load_traffic("grid_nearest_rec_kernel")
                call = [load_traffic(k)[0] for k in ("qp_hist_kernel", "qp_scatter_kernel", "grid_nearest_rec_kernel")]
                out[key]["call_traffic"] = sum(call) if all(c is not None for c in call) else None
                if out[key]["call_traffic"]:
                    out[key]["call_frac_traffic"] = out[key]["call_traffic"] / dt / 1e9 / HBM_PEAK_GBS
            else:
                traffic, src = load_traffic("grid_nearest_kernel<false>")
            out[key].update({"grid_kernel_ms": gms / gn,
                             "frac_survey_8d": ref / kernel_s / 1e9 / HBM_PEAK_GBS,  # > 1 possible: the grid reads less than the reference's walk
                             "frac_compulsory": (20 * len(c2q) + 16 * tree.Len()) / kernel_s / 1e9 / HBM_PEAK_GBS,
                             "traffic": traffic, "traffic_source": src,
                             "frac_traffic": traffic / kernel_s / 1e9 / HBM_PEAK_GBS if traffic else None})
        elif kn > 0:
            out[key]["frac_survey_8d"] = ref / (kms / kn * 1e-3) / 1e9 / HBM_PEAK_GBS
    out["knn_c2_unsorted"]["what"] = ("flags 0 of pcgx_kdtree_nearest_batch_dev: the queries are searched in the order they "
                                      "arrive -- the seam for batches that come spatially ordered (scan lines, the output of a "
                                      "voxel filter); on C2's random order it is the slower choice, and the host-pointer entry "
                                      "points never take it for 2^18 queries or more")
    # ... the case that seam is for: the same queries arriving ordered by cell (sorted once on the host here)
    cell = (np.floor(c2q / np.float32(0.625)).astype(np.int64) * np.array([1, 16, 256])).sum(axis=1)
    dqs = torch.from_numpy(np.ascontiguousarray(c2q[np.argsort(cell, kind="stable")])).to(dev)
    rows = {}
    for presort, key in ((False, "caller_order"), (True, "partitioned_again")):
        for _ in range(2):
            tree.NearestBatchDev(dqs.data_ptr(), len(c2q), 10.0, ids.data_ptr(), dsq.data_ptr(), presort, stream)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            tree.NearestBatchDev(dqs.data_ptr(), len(c2q), 10.0, ids.data_ptr(), dsq.data_ptr(), presort, stream)
        torch.cuda.synchronize()
        rows[key] = (time.perf_counter() - t0) / 20 * 1e3
    out["knn_c2_ordered_input"] = {"ms_per_call_caller_order": rows["caller_order"],
                                   "ms_per_call_partitioned_again": rows["partitioned_again"],
                                   "what": "C2's queries arriving sorted by 0.625 m cell: flags 0 against PCGX_KNN_PRESORT"}
    # Point-to-plane / Gauss-Newton extension (BASELINE.json config "ICP point-to-plane, 1M source vs
    # 1M target, 20 iters"; the reference has no such evaluator: no reference parity).
    cp = synth.c4_plane(1_000_000)
    ptree = kdtree.New(cp["base"])
    ps = icp.IcpSession(ptree, cp["target"], cp["max_dist"], cp["min_pairs"], None, cp["threshold"],
                        cp["max_iteration"], BaseNormals=cp["normals"])
    time_session_steps(torch, L, ps, 20, 20, stream)
    dt = time_session_steps(torch, L, ps, 60, 20, stream) / 60
    ptrans, pstat, _ = ps.result(stream)
    inv = np.linalg.inv(synth.icp_pose().astype(np.float64).reshape(4, 4).T).T.reshape(-1)
    out["icp_plane_c4"] = {"mpoints_per_s": len(cp["target"]) / dt / 1e6, "ms_per_step": dt * 1e3,
                           "final_value": float(pstat.Evaluated.Value),
                           "pose_error_max": float(np.max(np.abs(ptrans.astype(np.float64) - inv))),
                           "exchange_doubles": 30, "parity": "none in the reference (extension)"}
    ps.close()
    del ptree
    # Four independent Fits at once (the library's four pooled call contexts, pcgx_icp_fit from four host threads): the
    # latency-bound kernels of one Fit's sums run under the grid pass of another
    try:
        import threading
        c4 = synth.c4_icp()
        reg = icp.PointToPointICPGradient(
            icp.PointToPointEvaluator(icp.NearestPointCorresponder(MaxDist=c4["max_dist"]), MinPairs=c4["min_pairs"]),
            icp.GradientDescentUpdaterFactory(Weight=c4["weight"], Threshold=c4["threshold"], MaxIteration=c4["max_iteration"]))
        ctree = kdtree.New(c4["base"])
        reg.Fit(ctree, c4["target"])   # warm
        t0 = time.perf_counter()
        reg.Fit(ctree, c4["target"])
        one = time.perf_counter() - t0
        res = [None] * 4

        def worker(k):
            res[k] = reg.Fit(ctree, c4["target"])[0]
        for rep in range(3):   # (the first rounds grow the four contexts' workspaces; the last one is timed)
            th = [threading.Thread(target=worker, args=(k,)) for k in range(4)]
            t0 = time.perf_counter()
            for t in th:
                t.start()
            for t in th:
                t.join()
            four = time.perf_counter() - t0
        its = c4["max_iteration"]
        out["icp_c4_concurrent4"] = {
            "fits_in_flight": 4, "seconds_one_fit_alone": one, "seconds_four_fits": four,
            "gpoints_per_s_aggregate": 4 * its * len(c4["target"]) / four / 1e9,
            "gpoints_per_s_one_fit_alone": its * len(c4["target"]) / one / 1e9,
            "identical_results": bool(all(np.array_equal(res[0], r) for r in res[1:])),
            "note": "host-pointer Fits (12 MB target upload + session set-up inside each call), reference sums"}
        del ctree
    except Exception as e:  # noqa: BLE001
        out["icp_c4_concurrent4"] = {"error": str(e)[:200]}
    try:
        # the reference's own ICP benchmark (icp_test.go:100-142: a 10 x 10 m ground grid with a box, MinDistSq = res^2,
        # 10 iterations): one host-pointer Fit -- all iterations in ONE launch (csrc/icp_small.hip)
        small = {}
        for n_pts in (1024, 4096, 16384):
            width = int(np.sqrt(float(n_pts)))
            res = np.float32(10.0) / np.float32(width)
            i = np.arange(n_pts)
            bx = (res * (i // width).astype(np.float32) - np.float32(5)).astype(np.float32)
            by = (res * (i % width).astype(np.float32) - np.float32(5)).astype(np.float32)
            bz = np.where((bx > -1) & (bx < 1) & (by > -1) & (by < 1), np.float32(1), np.float32(0)).astype(np.float32)
            gbase = np.ascontiguousarray(np.stack([bx, by, bz], axis=1))
            gtarget = (gbase + np.array([0.5, 0.3, -0.2], np.float32)).astype(np.float32)
            gt = kdtree.New(gbase, MinDistSq=float(res * res))
            greg = icp.PointToPointICPGradient(
                icp.PointToPointEvaluator(icp.NearestPointCorresponder(MaxDist=2.0), MinPairs=3),
                icp.GradientDescentUpdaterFactory(Threshold=np.full(6, -1.0, np.float32), MaxIteration=10))
            best = 1e9
            for _ in range(6):
                t0 = time.perf_counter()
                greg.Fit(gt, gtarget)
                best = min(best, time.perf_counter() - t0)
            small[str(n_pts)] = best * 1e3
            del gt
        out["reference_icp_benchmark_fit_ms"] = small
    except Exception as e:  # noqa: BLE001
        out["reference_icp_benchmark_fit_ms"] = {"error": str(e)[:200]}
    c3 = synth.c3_voxel()
    dp = torch.from_numpy(c3["points"]).to(dev)
    dout = torch.empty_like(dp)
    for chunk, key in ((None, "voxel_c3"), ((64, 64, 64), "voxel_c3_chunked")):
        vg = voxelgrid.New(c3["leaf"]) if chunk is None else voxelgrid.New(c3["leaf"], voxelgrid.WithChunkSize(chunk))
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
        alg = 24 * n + 24 * m   # min/max pass + binning pass + representative record in and out
        traffic, src = load_traffic("voxel_pipeline")
        out[key] = {"mpoints_per_s": n / dt / 1e6, "ms_per_call": dt * 1e3, "out_points": int(m),
                    "algorithmic_gbs": alg / dt / 1e9, "roofline_frac": alg / dt / 1e9 / HBM_PEAK_GBS,
                    "frac_of_measured_copy_rate": alg / dt / 1e9 / hbm_gbs,
                    "traffic": traffic, "traffic_source": src,
                    "frac_traffic": traffic / dt / 1e9 / HBM_PEAK_GBS if traffic else None}
    return out


def launch_ranks(args):
    """`python bench.py --gpus N` outside torch.distributed.run: start the N ranks as a child (the
    parent never touches the GPU), pass its output through."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", choices=("c4", "c5"), default="c4")
    ap.add_argument("--points", type=int, default=0, help="base points (c4: also target points per GPU); 0: the config's size")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--f64-tree", action="store_true", help="N = 1: time the float64-tree reduction instead of the reference's sums")
    ap.add_argument("--no-reference-sharded", action="store_true", help="N > 1: skip the second timing (the float64-tree step beside the headline)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world

    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (no CPU fallback)")
    # Rehearsal of the N > 1 path on a one-GPU box: PCGX_BENCH_REHEARSE=1 puts every rank on cuda:0
    # and exchanges through the callback communicator over gloo (RCCL refuses two ranks on one
    # device).  Never used by the driver; such numbers are not comparable (ranks share the GPU).
    # =2: every rank on cuda:0 but RCCL is tried first -- it refuses, which exercises the fall-back below.
    rehearse = os.environ.get("PCGX_BENCH_REHEARSE") == "1"
    if os.environ.get("PCGX_BENCH_REHEARSE") in ("1", "2"):
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group("gloo")  # bootstrap, barriers, max over ranks; the data path's exchange is the library's

    from pcgol_amd import _lib as L
    from pcgol_amd import build, icp, kdtree, synth, voxelgrid
    from pcgol_amd.distributed import Comm
    L.check(L.lib().pcgx_init(local_rank))

    cfg = dict(max_dist=0.5, min_pairs=6, weight=np.full(6, 0.3, np.float32),
               threshold=np.full(6, -1.0, np.float32), max_iteration=20)
    if args.workload == "c5":
        # BASELINE config 5: ICP on a 64M-point cloud, the target tiled over 8 GPUs, the base tree replicated
        n = args.points or 64_000_000
        width = 40.0 * (n / 64e6) ** (1.0 / 3.0)   # the C4 point density
        base = synth.uniform_cloud_chunked(n, width, 2)
        c5_world = 8
        tile = synth.c5_tile(base, rank % c5_world, c5_world, width)
        workload = ("C5 ICP point-to-point gradient iteration on the %d-pt cloud tiled over 8 GPUs: %d-pt base KD-tree "
                    "(replicated), rank r holds octant r of the %d-pt target (synth.c5_tile, ~%d pts); %d of the 8 octants "
                    "are running; MaxDist 0.5, 20-iteration Fits, Threshold -1; one step = corr+reduce+re-projection+update"
                    % (n, n, n, n // c5_world, min(world, c5_world)))
    else:
        n = args.points or 1_000_000
        width = 10.0 * (n / 1e6) ** (1.0 / 3.0)  # keeps the C4 point density when --points is changed
        base = synth.uniform_cloud(n, width, 2)
        tile = synth.icp_tile(base, rank, world, n, width)  # this rank's spatial tile; nothing global is sorted
        workload = ("C4 ICP point-to-point gradient iteration (reference has no point-to-plane): %d-pt base KD-tree "
                    "(replicated) x ~%d target pts per GPU, MaxDist 0.5, 20-iteration Fits, Threshold -1; one step = "
                    "corr+reduce+re-projection+update" % (n, n))
    t0 = time.perf_counter()
    tree = kdtree.New(base)
    build_s = time.perf_counter() - t0

    side_stream = torch.cuda.Stream()  # a real (non-zero) stream handle
    stream = side_stream.cuda_stream
    # the library's default sums at every N: the reference's sequential float32 additions, over the ranks' tiles one
    # after the other when the target is spread over ranks (bit-identical to the Go code's Fit of the whole target)
    strict = not args.f64_tree
    sess = icp.IcpSession(tree, tile, cfg["max_dist"], cfg["min_pairs"], cfg["weight"], cfg["threshold"],
                          cfg["max_iteration"], SumsMode=icp.SumsReference if strict else icp.SumsF64Tree)
    comm = None
    exchange_fallback = None
    if world > 1:
        if rehearse:
            comm = Comm.gloo()
        else:
            # RCCL inside the library.  Rank 0 makes the id; whether that worked is agreed on BEFORE anybody
            # waits for the id (a rank 0 that cannot even load librccl must not leave the others in a broadcast),
            # and whether every rank's communicator came up is agreed on after: should either fail anywhere,
            # every rank takes the host callback over gloo instead and the line says so.
            import ctypes as C
            rccl_error = None
            idbuf = C.create_string_buffer(128)
            if rank == 0:
                try:
                    L.check(L.lib().pcgx_comm_unique_id(idbuf))
                except Exception as e:  # noqa: BLE001
                    rccl_error = str(e)
            msg = torch.zeros(129, dtype=torch.uint8)
            if rank == 0:
                msg[0] = 0 if rccl_error else 1
                msg[1:] = torch.tensor(list(idbuf.raw), dtype=torch.uint8)
            dist.broadcast(msg, 0)
            have_id = int(msg[0].item()) == 1
            if have_id:
                class Given:   # Comm.rccl's store: the id every rank already holds
                    def set(self, key, value):
                        pass

                    def get(self, key):
                        return bytes(msg[1:].tolist())
                try:
                    if rank == 0:
                        h = C.c_void_p()
                        L.check(L.lib().pcgx_comm_init(rank, world, idbuf, C.byref(h)))
                        comm = Comm(h, world=world)
                    else:
                        comm = Comm.rccl(rank, world, Given())
                except Exception as e:  # noqa: BLE001
                    rccl_error = str(e)
            elif rank != 0:
                rccl_error = "rank 0 could not make an RCCL id"
            ok = torch.tensor([0 if rccl_error else 1], dtype=torch.int32)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 0:
                if comm is not None:
                    comm.close()
                comm = Comm.gloo()
                exchange_fallback = "host callback over gloo (RCCL set-up failed on a rank%s)" % (
                    ": " + rccl_error[:120] if rccl_error else "")
    in_fit = [0]

    def step():
        if in_fit[0] == cfg["max_iteration"]:
            L.check(L.lib().pcgx_icp_session_reset(sess._h, L.ptr(stream)))
            in_fit[0] = 0
        if comm is None:
            sess.step(stream)  # grid pass -> leftover walk (+ tile sums) -> sums (reference: summaries, jobs, chain) + update
        else:
            L.check(L.lib().pcgx_icp_session_step_sharded(sess._h, comm._h, L.ptr(stream)))
        in_fit[0] += 1

    def barrier():
        if world > 1:
            dist.barrier()
        L.check(L.lib().pcgx_sync(L.ptr(stream)))
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    if world > 1 and strict:
        # The ring of the reference's sums has never run between two GPUs in this pipeline (no such box): if the warm-up
        # broke it on any rank -- a walker's wait ran out, an inbox could not be reached -- every rank takes the collective
        # form of the same sums instead (2 + N all-reduces per step, the same bits) through a fresh communicator, and
        # the line says so.  A line with the slower exchange is worth more than no line.
        broke = 0
        try:
            sess.result(stream)
        except Exception as e:  # noqa: BLE001
            broke = 1
            ring_error = str(e)[:160]
        if os.environ.get("PCGX_BENCH_TEST_BREAK_RING") and rank == world - 1:   # (tests: this path)
            broke, ring_error = 1, "PCGX_BENCH_TEST_BREAK_RING"
        flag = torch.tensor([broke], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        if int(flag.item()):
            os.environ["PCGX_SHARD_RING"] = "0"
            old_comm = comm
            comm = Comm.gloo() if (rehearse or exchange_fallback) else None
            if comm is None:
                try:   # a fresh RCCL communicator (the id from rank 0 again)
                    import ctypes as C
                    idbuf = C.create_string_buffer(128)
                    if rank == 0:
                        L.check(L.lib().pcgx_comm_unique_id(idbuf))
                    msg = torch.zeros(128, dtype=torch.uint8)
                    if rank == 0:
                        msg[:] = torch.tensor(list(idbuf.raw), dtype=torch.uint8)
                    dist.broadcast(msg, 0)
                    h = C.c_void_p()
                    L.check(L.lib().pcgx_comm_init(rank, world, C.create_string_buffer(bytes(msg.tolist()), 128), C.byref(h)))
                    comm = Comm(h, world=world)
                except Exception:  # noqa: BLE001
                    comm = Comm.gloo()
            old_comm.close()
            sess.close()
            sess = icp.IcpSession(tree, tile, cfg["max_dist"], cfg["min_pairs"], cfg["weight"], cfg["threshold"],
                                  cfg["max_iteration"], SumsMode=icp.SumsReference)
            in_fit[0] = 0
            exchange_fallback = (exchange_fallback + "; " if exchange_fallback else "") + \
                "the ring broke in the warm-up%s: collectives instead" % (": " + ring_error if broke else " on another rank")
            _st = np.zeros(4, np.int64)
            L.check(L.lib().pcgx_debug_shard_stats(L.ptr(_st), 1))   # (the line's shard_stats: the steps behind this point)
            for _ in range(args.warmup):
                step()
            barrier()
    L.prof_enable(0)   # nothing but the step's own launches inside the timed region
    # the driver's K may cover less than a millisecond of GPU work: repeat the K steps until >= 50 ms
    # have been timed and report per step (the JSON's `steps` stays the driver's K)
    rounds = 0
    elapsed = 0.0
    round_ms = []   # per step, of every timed round of K steps: the line's spread
    barrier()
    while True:
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        L.check(L.lib().pcgx_sync(L.ptr(stream)))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        elapsed += dt
        rounds += 1
        round_ms.append(dt / args.steps * 1e3)
        barrier()
        if elapsed >= 0.05 or rounds >= 200:
            break
    per_step = elapsed / (rounds * args.steps)
    trans, stat, _ = sess.result(stream)
    # ---- kernel times: a pass of its own (HIP events around every launch cost stream time), two Fits
    kinds = {"icp_grid_kernel": L.PROF_ICP_GRID, "icp_corr_kernel": L.PROF_ICP_WALK,
             "icp_corr_kernel (behind the grid pass)": L.PROF_ICP_LEFTOVER,
             "strict_tilesum_kernel": L.PROF_STRICT_TERMS, "strict_sum_kernel": L.PROF_STRICT_SUM,
             "strict_job_kernel": L.PROF_STRICT_JOB, "strict_chain_kernel": L.PROF_STRICT_CHAIN}
    L.check(L.lib().pcgx_icp_session_reset(sess._h, L.ptr(stream)))
    in_fit[0] = 0
    barrier()
    L.prof_enable(1)
    L.prof_reset()
    for _ in range(2 * cfg["max_iteration"]):
        step()
    barrier()
    kernel_ms = {}
    kernel_max_ms = {}
    kernel_launches = {}
    n_prof_steps = 2 * cfg["max_iteration"]
    for name, kind in kinds.items():
        ms, cnt = L.prof_read(kind)
        if cnt > 0:
            # per STEP: a kernel that is not launched in every step (the leftover walk behind the grid pass: a Fit's first
            # Evaluate only, csrc/icp.hip enqueue_corr) counts for what it costs a step on average
            kernel_ms[name] = ms / n_prof_steps
            kernel_max_ms[name] = L.prof_read_max(kind)
            kernel_launches[name] = cnt / n_prof_steps
    L.prof_enable(0)
    n_tile = len(tile)
    n_total = n_tile
    if world > 1:
        t = torch.tensor([n_tile], dtype=torch.int64)
        dist.all_reduce(t)
        n_total = int(t.item())
    # the same GPU in the float64-tree mode without any exchange: N = 1's `value_f64_tree`, N > 1's `value_same_mode_n1`
    if strict:
        s64 = icp.IcpSession(tree, tile, cfg["max_dist"], cfg["min_pairs"], cfg["weight"], cfg["threshold"],
                             cfg["max_iteration"], SumsMode=icp.SumsF64Tree)
    else:
        s64 = sess
    time_session_steps(torch, L, s64, 20, 20, stream)
    t64 = time_session_steps(torch, L, s64, 60, 20, stream) / 60
    if world > 1:
        t = torch.tensor([t64], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        t64 = float(t.item())
    f64_one_gpu = n_tile / t64 / 1e6
    # N > 1, beside the headline: the same sharded step with float64 sums (PCGX_SUMS_F64_TREE: ONE all-reduce of ten
    # doubles per step; differs from the reference by the reference's own rounding noise) -- N = 1's `value_f64_tree`
    f64_sharded = None
    if world > 1 and comm is not None and strict and not args.no_reference_sharded:
        try:
            def f64_steps(k):
                for i in range(k):
                    if i % cfg["max_iteration"] == 0:
                        L.check(L.lib().pcgx_icp_session_reset(s64._h, L.ptr(stream)))
                    L.check(L.lib().pcgx_icp_session_step_sharded(s64._h, comm._h, L.ptr(stream)))
            f64_steps(cfg["max_iteration"])
            barrier()
            t0 = time.perf_counter()
            f64_steps(2 * cfg["max_iteration"])
            L.check(L.lib().pcgx_sync(L.ptr(stream)))
            torch.cuda.synchronize()
            tr = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
            dist.all_reduce(tr, op=dist.ReduceOp.MAX)
            f64_sharded = float(tr.item()) / (2 * cfg["max_iteration"])
            barrier()
        except Exception as e:  # noqa: BLE001  (never at the price of the headline line)
            f64_sharded = "failed: %s" % str(e)[:160]
    if s64 is not sess:
        s64.close()
    shard_stats = np.zeros(4, np.int64)
    L.check(L.lib().pcgx_debug_shard_stats(L.ptr(shard_stats), 0))
    ring_kinds = np.zeros(2, np.int64)   # rank 0's count: {rings with the inboxes in the ranks' device memory, in host memory}
    L.check(L.lib().pcgx_debug_ring_kinds(L.ptr(ring_kinds), 0))

    # What the grid pass reads per iteration (an untimed Fit, one instrumented launch before each step)
    grid_pts = grid_words = grid_walked = grid_kept = 0
    have_grid = "icp_grid_kernel" in kernel_ms
    if have_grid:
        L.check(L.lib().pcgx_icp_session_reset(sess._h, L.ptr(stream)))
        in_fit[0] = 0
        for _ in range(cfg["max_iteration"]):
            g = sess.grid_stats(stream)
            grid_walked += g[1]
            grid_pts += g[2]
            grid_words += g[3]
            grid_kept += g[5]
            step()
        barrier()

    if rank == 0:
        visits = load_visits()
        hbm_gbs = hbm_copy_gbs(torch)
        v_icp = visits["c4_icp"]["mean_visits_per_point"]
        its = cfg["max_iteration"]
        summary, traffic_src, same_build = load_pmc_summary()
        usable = same_build and args.workload == "c4" and n == 1_000_000 and world == 1
        # ---- the step, kernel by kernel: time (this run), HBM traffic (PMC summary of the same command)
        kernels = {}
        step_traffic = 0.0
        traffic_complete = usable
        for name, ms in kernel_ms.items():
            key = name.split(" ")[0]
            if key == "icp_grid_kernel":   # the instantiation this session launches (reference sums: no sums in the grid pass)
                key = "icp_grid_kernel<false, false, false>" if strict else "icp_grid_kernel<false, false, true>"
            if "behind the grid pass" in name:
                key = "icp_corr_kernel<false, false, true, false>" if strict else "icp_corr_kernel<false, false, true, true>"
            tb = traffic_of(summary, key) if usable else None
            if tb and kernel_launches.get(name, 1.0) < 1.0:
                tb *= kernel_launches[name]   # (the PMC summary is per launch)
            kernels[name] = {"ms": ms, "launches_per_step": kernel_launches.get(name, 1.0), "traffic": tb,
                             "frac": tb / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS if tb else None}
            if tb:
                step_traffic += tb
            else:
                traffic_complete = False
        survey_bytes = (12 + 16 * v_icp) * n_tile       # 12 B target + 16 B per node the REFERENCE walk touches
        # what any exact method must move per step: the target, the previous pair in / the new pair out
        # (iterations >= 1 read one), and the base points once
        compulsory = (12 + 16 * (its - 1) / its + 16) * n_tile + 16 * n
        longest = max(kernel_ms, key=kernel_ms.get)
        step_s = per_step
        frac_step_traffic = step_traffic / step_s / 1e9 / HBM_PEAK_GBS if traffic_complete else None
        frac_step_compulsory = compulsory / step_s / 1e9 / HBM_PEAK_GBS
        roof = {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s",
                # the roofline of the STEP: bytes the step's kernels moved (PMC, summed) / the driver-timed step / peak;
                # without a PMC summary of this build: the compulsory bytes
                "kernel": "step (all kernels of one ICP iteration; the longest is %s)" % longest,
                "traffic": step_traffic if traffic_complete else None,
                "traffic_source": traffic_src, "traffic_is_of_this_build": bool(same_build),
                "achieved": (step_traffic if traffic_complete else compulsory) / step_s / 1e9,
                "frac": frac_step_traffic if frac_step_traffic is not None else frac_step_compulsory,
                "frac_basis": "PMC traffic of the step's kernels" if frac_step_traffic is not None else
                              "compulsory bytes (no PMC summary of this build and size)",
                "step": {"ms": per_step * 1e3, "traffic": step_traffic if traffic_complete else None,
                         "frac_traffic": frac_step_traffic, "frac_compulsory": frac_step_compulsory,
                         # the step priced as SURVEY 8(d) prices the reference's walk; the certified grid pass answers with
                         # fewer bytes than that walk would stream -- an algorithm change (identical results), not skipped work
                         "frac_survey_8d": survey_bytes / step_s / 1e9 / HBM_PEAK_GBS,
                         "survey_8d_bytes": survey_bytes, "compulsory_bytes": compulsory,
                         "ms_sum_of_kernels": sum(kernel_ms.values()), "kernels": kernels},
                "measured_hbm_copy_gbs": hbm_gbs,
                "frac_of_measured_copy_rate": (step_traffic if traffic_complete else compulsory) / step_s / 1e9 / hbm_gbs,
                "note": "the C4 working set (~62 MB) sits inside the 256 MB Infinity Cache: 'HBM' is nominal for every "
                        "kernel of this step; the sums' kernels are latency-bound (8 workgroups walk the chain)"}
        if have_grid:
            v_pts, v_words = grid_pts / (its * n_tile), grid_words / (its * n_tile)
            kept_share = grid_kept / (its * n_tile)
            roof["grid_pass"] = {"point_records_per_searched_target": v_pts, "bound_words_per_searched_target": v_words,
                                 "targets_left_to_walk_per_fit": grid_walked,
                                 "share_of_targets_kept_on_a_certificate_without_a_search": kept_share,
                                 "bytes_the_kernel_chooses_to_read": (12 + 20 * (its - 1) / its + 16
                                                                      + (1.0 - kept_share) * (16 * v_pts + 4 * v_words)) * n_tile,
                                 "what": "a target whose DistSq to last iteration's partner is below the partner's certificate keeps "
                                         "it (csrc/knn_grid.hip, grid_cert_kernel); the records and words are those of the targets "
                                         "that are searched for"}
        mode = ("reference: the evaluator's sequential float32 sums, bit-identical to the Go code (the library's default; "
                "tests/test_gpu_icp.py)"
                if strict else "f64-tree: float64 reduction of the reference's float32 terms (differs from the Go code by "
                               "its own rounding noise, 1.6e-5 on the final transform at 1M pairs)")
        line = {
            "metric": "Mpoints/sec ICP iter (corr+reduce) + kNN queries/sec, 1M-pt cloud",
            "value": n_total / per_step / 1e6,
            "unit": "Mpoints/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": per_step * 1e3,
            # the spread of the timed rounds (each K steps, clocked by itself), and the slowest single launch of the
            # chain kernel in two profiled Fits (the iterations in which a sum hovers around zero)
            "ms_per_step_min": min(round_ms), "ms_per_step_median": float(np.median(round_ms)), "ms_per_step_max": max(round_ms),
            "worst_iteration_us": {k.split(" ")[0]: v * 1e3 for k, v in kernel_max_ms.items() if k.startswith("strict_")},
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "parity_mode": mode,
            "timed_rounds_of_steps": rounds,
            "config": {"workload": workload,
                       "base_points": n, "target_points_total": n_total,
                       "parallelism": "spatial target tiles x%d (%s), tree replicated" % (
                           world, "synth.c5_tile: octants" if args.workload == "c5" else "synth.spatial_cell"),
                       "exchange": "none" if world == 1 else (
                           ("%s; communicator: %s" % (
                               ("reference sums over the ranks: ring of tagged words, no collective per step (csrc/strict.hip "
                                "strict_enqueue_ring); inboxes in %s" % (
                                    "the ranks' DEVICE memory, mapped by the peers through HIP IPC handles (a hop: one store over "
                                    "xGMI, a poll of local HBM)" if ring_kinds[0] > 0 else
                                    "shared HOST memory (a rank could not export or map a device inbox: a PCIe round trip per poll)")
                                if shard_stats[0] > 0 and shard_stats[1] == 0 else
                                "reference sums over the ranks: 2 + N all-reduces per step (no shared-memory ring between these ranks)")
                               if strict else "all-reduce 10 x f64 per step",
                               "callback over gloo: REHEARSAL on one GPU, not a measurement" if rehearse
                               else (exchange_fallback or "RCCL inside libpcgx.so"))))},
            "roofline": roof,
            "tree_build_s": build_s,
            "final_value": float(stat.Evaluated.Value),
            "source_hash": build.source_hash(),
        }
        if world > 1:
            if exchange_fallback:
                line["exchange_note"] = exchange_fallback
            line["shard_stats"] = {"ring_steps": int(shard_stats[0]), "collective_steps": int(shard_stats[1]),
                                   "rings_made": int(shard_stats[2]), "ring_setups_fallen_back": int(shard_stats[3]),
                                   "rings_in_device_memory": int(ring_kinds[0]), "rings_in_host_memory": int(ring_kinds[1])}
            # one of these GPUs alone in the float64 mode, no exchange
            line["value_f64_tree_one_gpu_alone"] = f64_one_gpu
            if isinstance(f64_sharded, float):
                line["value_f64_tree"] = n_total / f64_sharded / 1e6
                line["ms_per_step_f64_tree"] = f64_sharded * 1e3
            elif f64_sharded is not None:
                line["value_f64_tree"] = None
                line["f64_tree_note"] = f64_sharded
            if strict:   # (the name earlier rounds' lines carried the reference-sums step under)
                line["ms_per_step_reference_sums"] = per_step * 1e3
        else:
            line["value_f64_tree"] = f64_one_gpu
        if world == 1 and not args.no_extras and args.workload == "c4":
            line["extra"] = side_benchmarks(torch, L, kdtree, voxelgrid, synth, tree, stream, visits, icp, hbm_gbs)
            # the second half of BASELINE.json's metric ("+ kNN queries/sec, 1M-pt cloud", config C2)
            line["knn_queries_per_s"] = line["extra"]["knn_c2_presort"]["mqueries_per_s"] * 1e6
            c4c = line["extra"].get("icp_c4_concurrent4", {})
            if "seconds_one_fit_alone" in c4c:
                # what PointToPointICPGradient.Fit(base, target) (icp.go:23) costs a caller that holds host slices: the
                # 12 MB target up, session set-up, twenty iterations, the result back -- PCIe included, never `value`
                line["fit_host_pointer_ms"] = c4c["seconds_one_fit_alone"] * 1e3
                line["four_fits_in_flight_ms"] = c4c["seconds_four_fits"] * 1e3
        if world == 1 and not args.no_cpu_baseline and args.workload == "c4":
            line["cpu_baseline"] = cpu_baseline(synth, base, tile, cfg)   # (C5: the oracle's tree build alone takes minutes)
        print(json.dumps(line), flush=True)
    sess.close()
    if comm is not None:
        comm.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
