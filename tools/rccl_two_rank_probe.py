"""The one component no test here has ever executed with more than one rank: the library's RCCL exchange
(csrc/comm.hip: ncclCommInitRank / ncclAllReduce bound at run time), because RCCL refuses two ranks on one
device and a gpurun box has one GPU.  For the first multi-GPU lease:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 \
        tools/rccl_two_rank_probe.py            (or any N <= the node's GPUs)

Every rank: Comm.rccl over a gloo bootstrap, then 1000 all-reduces of 10 float64 (the per-iteration exchange of
SURVEY 8(e)) on its own GPU.  Checks: the sum is the exact expected one, bitwise the same on every rank; prints
microseconds per exchange (stream-ordered, no host synchronisation in between) and a short sharded Fit whose
transform must be identical on all ranks."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dist.init_process_group("gloo")
    from pcgol_amd import _lib as L
    from pcgol_amd import kdtree, synth
    from pcgol_amd.distributed import Comm, ShardedIcp
    L.check(L.lib().pcgx_init(local))

    class Store:  # the 128-byte id travels over the gloo group
        def set(self, key, value):
            dist.broadcast(torch.tensor(list(value), dtype=torch.uint8), 0)

        def get(self, key):
            t = torch.zeros(128, dtype=torch.uint8)
            dist.broadcast(t, 0)
            return bytes(t.tolist())
    comm = Comm.rccl(rank, world, Store())
    stream = torch.cuda.Stream()
    buf = torch.zeros(10, dtype=torch.float64, device="cuda")
    reps = 1000
    expect = np.zeros(10)
    with torch.cuda.stream(stream):
        for it in range(reps + 10):
            if it == 10:
                stream.synchronize()
                dist.barrier()
                t0 = time.perf_counter()
            # every rank contributes (rank + 1) * 2^-k in slot k: sums of few-bit numbers, exact in any order
            buf.copy_(torch.tensor([(rank + 1) * 2.0 ** -k for k in range(10)], dtype=torch.float64), non_blocking=False)
            L.check(L.lib().pcgx_comm_allreduce_f64(comm._h, L.ptr(buf.data_ptr()), 10, L.ptr(stream.cuda_stream)))
        stream.synchronize()
    dt = (time.perf_counter() - t0) / reps
    expect = np.array([world * (world + 1) / 2 * 2.0 ** -k for k in range(10)])
    got = buf.cpu().numpy()
    ok = np.array_equal(got, expect)
    allg = [torch.zeros(10, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(allg, torch.from_numpy(got))
    same = all(np.array_equal(a.numpy().view(np.uint64), got.view(np.uint64)) for a in allg)
    # a short sharded Fit: every rank must end with the same transform
    c = synth.c4_icp(n=200_000, width=10.0 * 0.2 ** (1 / 3))
    cell = synth.spatial_cell(c["target"], world, (0, 0, 0), (10.0 * 0.2 ** (1 / 3),) * 3)
    tile = np.ascontiguousarray(c["target"][cell == rank])
    s = ShardedIcp(kdtree.New(c["base"]), tile, c["max_dist"], c["min_pairs"], c["weight"], c["threshold"], c["max_iteration"],
                   comm=comm)
    tr, st, _ = s.fit()
    s.close()
    trs = [torch.zeros(16, dtype=torch.float32) for _ in range(world)]
    dist.all_gather(trs, torch.from_numpy(tr))
    fit_same = all(np.array_equal(a.numpy(), tr) for a in trs)
    if rank == 0:
        print("rccl exchange over %d ranks: sum exact %s, bitwise equal on all ranks %s, %.1f us per all-reduce of 10 x f64 "
              "(incl. the 80-byte refill copy); sharded Fit identical on all ranks: %s, %d iterations" % (
                  world, ok, same, dt * 1e6, fit_same, st.NumIteration))
    comm.close()
    dist.barrier()
    dist.destroy_process_group()
    if not (ok and same and fit_same):
        sys.exit(1)


if __name__ == "__main__":
    main()
