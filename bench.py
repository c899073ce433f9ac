#!/usr/bin/env python3
"""bench.py -- headline benchmark of the pcgol_amd hot path on MI355X.

Metric (BASELINE.json): Mpoints/s of one ICP iteration (correspondence + reduction [+ re-projection
+ pose update]) on the 1M-point cloud, with the kNN queries/s (C2) and VoxelGrid Mpoints/s (C3)
figures measured alongside.

A "step" is ONE ICP iteration over this rank's tile of the target (1M points per GPU: weak scaling)
against the replicated 1M-point base KD-tree.  Every 20 steps a new Fit starts (state reset), as in
the reference's MaxIteration = 20 loop (icp.go:48-65); Threshold = -1 keeps all iterations.

  N = 1   the path that MATCHES THE REFERENCE BIT FOR BIT ("parity_mode": "strict"): certified
          grid pass -> leftover walk -> the evaluator's sequential float32 sums evaluated exactly in
          parallel (csrc/strict_sum.h) -> pose update, all enqueued on the device.
  N > 1   one spatial tile of the target per GPU, 10 float64 partial sums, ONE all-reduce per step
          through the library's own RCCL communicator (pcgx_icp_session_step_sharded).  A sum
          spread over ranks has no sequential order: "parity_mode": "f64-tree" (float64 reduction of
          the reference's float32 terms; differs from the Go code by ITS rounding noise).  The
          N = 1 line carries the same-mode single-GPU figure in extra.icp_f64_tree_c4.

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


def load_traffic(kernel):
    """HBM bytes per launch of `kernel` from the newest committed PMC summary (profiles/rNN*_pmc.json,
    written by profiles/collect.sh: FETCH_SIZE and WRITE_SIZE collected in separate --pmc passes of this
    same bench command).  rocprofv3 reports both in KiB; FETCH_SIZE is doubled for wide streaming reads
    only (gfx950 tallies their 128-byte requests at 64 bytes, MI355X_MICROARCH.md); a random gather is
    reported at the sector bytes it costs (profiles/rNN_fetch_probe.json: the check on a streaming
    copy, a 16-byte and a 4-byte gather), so gather kernels take the counter as it is plus half of
    their known coalesced reads (summarize.py); WRITE_SIZE is exact.  PMC collection cannot run inside the timed process, hence the
    committed summary."""
    import glob
    for p in reversed(sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9]*_pmc.json")))):
        try:
            with open(p) as f:
                d = json.load(f)
        except (OSError, ValueError):
            continue
        for name, c in sorted(d.items(), key=lambda kv: -len(kv[0])):  # the most specific entry first
            if kernel in name and "FETCH_SIZE" in c and "WRITE_SIZE" in c:
                scale = float(c.get("fetch_scale", 2.0))
                fetch = c["FETCH_SIZE"]["mean_per_dispatch"] * 1024.0 * scale + float(c.get("fetch_add_bytes", 0.0))
                write = c["WRITE_SIZE"]["mean_per_dispatch"] * 1024.0
                return fetch + write, os.path.basename(p)
    return None, None


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
            traffic, src = load_traffic("grid_nearest_kernel<false> [queries in %s order]" % ("Morton" if presort else "caller"))
            if traffic is None:
                traffic, src = load_traffic("grid_nearest_kernel")
            out[key].update({"grid_kernel_ms": gms / gn,
                             "frac_survey_8d": ref / kernel_s / 1e9 / HBM_PEAK_GBS,  # > 1 possible: the grid reads less than the reference's walk
                             "frac_compulsory": (20 * len(c2q) + 16 * tree.Len()) / kernel_s / 1e9 / HBM_PEAK_GBS,
                             "traffic": traffic, "traffic_source": src,
                             "frac_traffic": traffic / kernel_s / 1e9 / HBM_PEAK_GBS if traffic else None})
        elif kn > 0:
            out[key]["frac_survey_8d"] = ref / (kms / kn * 1e-3) / 1e9 / HBM_PEAK_GBS
    # the float64-tree reduction mode (round 1's default; what the sharded path computes per GPU)
    c4 = synth.c4_icp()
    s0 = icp.IcpSession(tree, c4["target"], c4["max_dist"], c4["min_pairs"], c4["weight"], c4["threshold"],
                        c4["max_iteration"], SumsMode=icp.SumsF64Tree)
    time_session_steps(torch, L, s0, 20, 20, stream)
    dt = time_session_steps(torch, L, s0, 100, 20, stream) / 100
    out["icp_f64_tree_c4"] = {"mpoints_per_s": len(c4["target"]) / dt / 1e6, "ms_per_step": dt * 1e3,
                              "parity": "float64 reduction of the reference's float32 terms: 1.6e-5 from the Go-semantics "
                                        "oracle on the final transform at this size (the reference's own rounding noise)"}
    s0.close()
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
    ap.add_argument("--points", type=int, default=1_000_000, help="target points per GPU (and base size)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--f64-tree", action="store_true", help="N = 1: time the float64-tree reduction instead of the strict sums")
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
    from pcgol_amd import icp, kdtree, synth, voxelgrid
    from pcgol_amd.distributed import Comm
    L.check(L.lib().pcgx_init(local_rank))

    n = args.points
    width = 10.0 * (n / 1e6) ** (1.0 / 3.0)  # keeps the C4 point density when --points is changed
    base = synth.uniform_cloud(n, width, 2)
    cfg = dict(max_dist=0.5, min_pairs=6, weight=np.full(6, 0.3, np.float32),
               threshold=np.full(6, -1.0, np.float32), max_iteration=20)
    tile = synth.icp_tile(base, rank, world, n, width)  # this rank's spatial tile; nothing global is sorted
    t0 = time.perf_counter()
    tree = kdtree.New(base)
    build_s = time.perf_counter() - t0

    side_stream = torch.cuda.Stream()  # a real (non-zero) stream handle
    stream = side_stream.cuda_stream
    strict = world == 1 and not args.f64_tree
    sess = icp.IcpSession(tree, tile, cfg["max_dist"], cfg["min_pairs"], cfg["weight"], cfg["threshold"],
                          cfg["max_iteration"], SumsMode=icp.SumsReference if strict else icp.SumsF64Tree)
    comm = None
    exchange_fallback = None
    if world > 1:
        if rehearse:
            comm = Comm.gloo()
        else:
            class BroadcastStore:   # the ncclUniqueId travels over the gloo group
                def set(self, key, value):
                    dist.broadcast(torch.tensor(list(value), dtype=torch.uint8), 0)

                def get(self, key):
                    t = torch.zeros(128, dtype=torch.uint8)
                    dist.broadcast(t, 0)
                    return bytes(t.tolist())
            # RCCL inside the library; should its set-up fail on any rank (missing library, IPC refused),
            # every rank takes the host callback over gloo instead and the line says so
            rccl_error = None
            try:
                comm = Comm.rccl(rank, world, BroadcastStore())
            except Exception as e:  # noqa: BLE001
                rccl_error = str(e)
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
            sess.step(stream)  # grid pass -> leftover walk -> sums (strict: terms, summaries, chain) + update
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
    # HIP events around every 7th launch of the timed kernels only (7 is coprime to the 20 iterations
    # of a Fit, so every iteration index is sampled): a pair of events costs ~5 us of stream time
    L.prof_enable(7)
    L.prof_reset()
    # the driver's K may cover less than a millisecond of GPU work: repeat the K steps until >= 50 ms
    # have been timed and report per step (the JSON's `steps` stays the driver's K)
    rounds = 0
    elapsed = 0.0
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
        barrier()
        if elapsed >= 0.05 or rounds >= 200:
            break
    per_step = elapsed / (rounds * args.steps)
    kinds = {"icp_grid_kernel": L.PROF_ICP_GRID, "icp_corr_kernel": L.PROF_ICP_WALK, "strict_terms_kernel": L.PROF_STRICT_TERMS,
             "strict_sum_kernel": L.PROF_STRICT_SUM, "strict_chain_kernel": L.PROF_STRICT_CHAIN}
    kernel_ms = {}
    for name, kind in kinds.items():
        ms, cnt = L.prof_read(kind)
        if cnt > 0:
            kernel_ms[name] = ms / cnt
    L.prof_enable(0)
    trans, stat, _ = sess.result(stream)
    n_tile = len(tile)
    n_total = n_tile
    if world > 1:
        t = torch.tensor([n_tile], dtype=torch.int64)
        dist.all_reduce(t)
        n_total = int(t.item())
        # the same GPUs without the exchange (each its own tile): reference for the scaling efficiency
        t0 = time_session_steps(torch, L, sess, 40, 20, stream) / 40
        t = torch.tensor([t0], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        no_exchange = n_total / float(t.item()) / 1e6

    # What the grid pass reads per iteration (an untimed Fit, one instrumented launch before each step)
    grid_pts = grid_words = grid_walked = 0
    have_grid = "icp_grid_kernel" in kernel_ms
    if have_grid:
        L.check(L.lib().pcgx_icp_session_reset(sess._h, L.ptr(stream)))
        for _ in range(cfg["max_iteration"]):
            g = sess.grid_stats(stream)
            grid_walked += g[1]
            grid_pts += g[2]
            grid_words += g[3]
            step() if comm is not None else sess.step(stream)
        barrier()

    if rank == 0:
        visits = load_visits()
        hbm_gbs = hbm_copy_gbs(torch)
        v_icp = visits["c4_icp"]["mean_visits_per_point"]
        its = cfg["max_iteration"]
        kernel = "icp_grid_kernel" if have_grid else "icp_corr_kernel"
        kernel_s = kernel_ms[kernel] * 1e-3
        survey_bytes = (12 + 16 * v_icp) * n_tile       # 12 B target + 16 B per node the REFERENCE walk touches
        # what any exact method must move per launch: the target, the previous pair in / the new pair
        # out (iterations >= 1 read one), and the base points once
        compulsory = (12 + 16 * (its - 1) / its + 16) * n_tile + 16 * n
        traffic, traffic_src = load_traffic(kernel) if n == 1_000_000 else (None, None)
        frac_traffic = traffic / kernel_s / 1e9 / HBM_PEAK_GBS if traffic else None
        roof = {"bound": "hbm", "kernel": kernel,
                "kernel_role": "correspondence pass: the kernel that moves the bytes of a step (the strict chain "
                               "kernel is a latency-bound scalar walk over ~300 KB of tile records; see `kernels_ms`)",
                "kernel_ms": kernel_ms[kernel], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "traffic": traffic, "traffic_source": traffic_src,
                # frac = measured HBM bytes (PMC) / kernel time / peak; falls back to the compulsory bytes without a PMC summary
                "achieved": (traffic if traffic else compulsory) / kernel_s / 1e9,
                "frac": frac_traffic if frac_traffic is not None else compulsory / kernel_s / 1e9 / HBM_PEAK_GBS,
                "frac_traffic": frac_traffic,
                "frac_compulsory": compulsory / kernel_s / 1e9 / HBM_PEAK_GBS,
                # the same launch priced as SURVEY 8(d) prices the reference's walk; above 1: the certified
                # grid pass answers with fewer bytes than that walk would stream -- an algorithm change
                # (identical results), not skipped work
                "frac_survey_8d": survey_bytes / kernel_s / 1e9 / HBM_PEAK_GBS,
                "survey_8d_bytes_per_launch": survey_bytes, "compulsory_bytes_per_launch": compulsory,
                "measured_hbm_copy_gbs": hbm_gbs,
                "frac_of_measured_copy_rate": (traffic if traffic else compulsory) / kernel_s / 1e9 / hbm_gbs,
                "kernels_ms": kernel_ms,
                "step_ms_sum_of_kernels": sum(kernel_ms.values())}
        if have_grid:
            v_pts, v_words = grid_pts / (its * n_tile), grid_words / (its * n_tile)
            roof.update({"point_records_per_target": v_pts, "bound_words_per_target": v_words,
                         "targets_left_to_walk_per_fit": grid_walked,
                         "bytes_the_kernel_chooses_to_read": (12 + 16 * (its - 1) / its + 16 + 16 * v_pts + 4 * v_words) * n_tile})
        mode = ("strict: the evaluator's sequential float32 sums, bit-identical to the Go code (tests/test_gpu_icp.py)"
                if strict else "f64-tree: float64 reduction of the reference's float32 terms (differs from the Go code by "
                               "its own rounding noise, 1.6e-5 on the final transform at 1M pairs; a sum spread over "
                               "ranks has no sequential order to reproduce)")
        line = {
            "metric": "Mpoints/sec ICP iter (corr+reduce) + kNN queries/sec, 1M-pt cloud",
            "value": n_total / per_step / 1e6,
            "unit": "Mpoints/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": per_step * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "parity_mode": mode,
            "timed_rounds_of_steps": rounds,
            "config": {"workload": "C4 ICP point-to-point gradient iteration (reference has no point-to-plane): "
                                   "%d-pt base KD-tree (replicated) x ~%d target pts per GPU, MaxDist 0.5, "
                                   "20-iteration Fits, Threshold -1; one step = corr+reduce+re-projection+update"
                                   % (n, n),
                       "base_points": n, "target_points_total": n_total,
                       "parallelism": "spatial target tiles x%d (synth.spatial_cell), tree replicated" % world,
                       "exchange": "none" if world == 1 else "all-reduce 10 x f64 per step (%s)"
                                   % ("callback over gloo: REHEARSAL on one GPU, not a measurement" if rehearse
                                      else (exchange_fallback or "RCCL inside libpcgx.so"))},
            "roofline": roof,
            "tree_build_s": build_s,
            "final_value": float(stat.Evaluated.Value),
        }
        if world > 1:
            line["scaling_reference"] = {"same_gpus_no_exchange_mpoints_per_s": no_exchange,
                                         "note": "N = 1 times the strict sums; compare N > 1 with this figure or with the "
                                                 "N = 1 line's extra.icp_f64_tree_c4"}
        if world == 1 and not args.no_extras:
            line["extra"] = side_benchmarks(torch, L, kdtree, voxelgrid, synth, tree, stream, visits, icp, hbm_gbs)
            # the second half of BASELINE.json's metric ("+ kNN queries/sec, 1M-pt cloud", config C2)
            line["knn_queries_per_s"] = line["extra"]["knn_c2_presort"]["mqueries_per_s"] * 1e6
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(synth, base, tile, cfg)
        print(json.dumps(line), flush=True)
    sess.close()
    if comm is not None:
        comm.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
