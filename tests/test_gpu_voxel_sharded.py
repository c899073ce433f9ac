"""The voxel filter over several ranks (pcgx_voxel_filter_sharded_dev; SURVEY 8(e), second half:
min/max exchange + each rank its contiguous share of the reference's output order).

Three PROCESSES on the one GPU of the test box, each holding the same cloud, exchanging through the
callback communicator (gloo underneath: RCCL refuses several ranks on one device).  The ranks'
outputs, rank 0's first, must BE the one-GPU filter's output byte for byte -- plain mode, chunked
with the combined key, chunked with the two sorts, a record with a label carried along -- and the
one-GPU output is the oracle's (tests/test_gpu_voxel.py)."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORLD = 3


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _cases():
    from pcgol_amd import synth
    pts = synth.uniform_cloud(300_000, 1.6, 11)
    shifted = (pts + np.float32(0.4)).astype(np.float32)  # vMin != 0 (plain mode sizes its grid by vMax: voxelgrid.go:46)
    rec = np.zeros((len(pts), 4), np.float32)             # x y z label, stride 16
    rec[:, :3] = pts
    rec[:, 3] = np.arange(len(pts), dtype=np.float32)
    # a cloud that leaves most of the key range empty: two clusters in opposite corners
    two = np.concatenate([synth.uniform_cloud(50_000, 0.2, 12),
                          (synth.uniform_cloud(50_000, 0.2, 13) + np.float32(1.4)).astype(np.float32)])
    return [
        ("plain", pts, 12, (0.05, 0.05, 0.05), None, None),
        ("plain shifted", shifted, 12, (0.05, 0.05, 0.05), None, None),
        ("chunked, one key", pts, 12, (0.05, 0.05, 0.05), (8, 8, 8), None),
        ("chunked, two sorts", pts, 12, (0.05, 0.05, 0.05), (8, 8, 8), "1"),
        ("labelled records", rec, 16, (0.04, 0.05, 0.06), (16, 16, 16), None),
        ("two clusters", two, 12, (0.01, 0.01, 0.01), None, None),
        ("fewer points than ranks' slices need", pts[:2], 12, (0.05, 0.05, 0.05), None, None),
    ]


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pcgol_amd import voxelgrid
        from pcgol_amd.distributed import Comm
        comm = Comm.gloo()
        out = []
        for name, data, stride, leaf, chunk, two in _cases():
            if two:
                os.environ["PCGX_VOXEL_TWO_SORTS"] = two
            else:
                os.environ.pop("PCGX_VOXEL_TWO_SORTS", None)
            vg = voxelgrid.New(leaf) if chunk is None else voxelgrid.New(leaf, voxelgrid.WithChunkSize(chunk))
            d = torch.from_numpy(np.ascontiguousarray(data)).cuda()
            o = torch.empty_like(d)
            m = vg.FilterShardDev(comm, d.data_ptr(), len(data), stride, 0, o.data_ptr())
            torch.cuda.synchronize()
            out.append((name, m, o.cpu().numpy().reshape(len(data), -1)[:m].copy()))
        # host buffers in and out (what the Go shim's FilterSharded calls)
        part = voxelgrid.New((0.05, 0.05, 0.05)).FilterShard(_cases()[0][1][:50_000], comm)
        out.append(("host", part.Points, np.frombuffer(part.Data.tobytes(), np.float32).reshape(-1, 3).copy()))
        comm.close()
        q.put((rank, out))
    except Exception as e:  # the parent must not wait for a rank that died
        import traceback
        q.put((rank, "rank %d: %s\n%s" % (rank, e, traceback.format_exc())))
    finally:
        dist.destroy_process_group()


def test_ranks_outputs_concatenated_are_the_one_gpu_output():
    import torch
    import torch.multiprocessing as mp
    from pcgol_amd import voxelgrid
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, WORLD, port, q)) for r in range(WORLD)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=150) for _ in range(WORLD)), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
    for r in res:
        assert not isinstance(r[1], str), r[1]
    os.environ.pop("PCGX_VOXEL_TWO_SORTS", None)
    for k, (name, data, stride, leaf, chunk, two) in enumerate(_cases()):
        vg = voxelgrid.New(leaf) if chunk is None else voxelgrid.New(leaf, voxelgrid.WithChunkSize(chunk))
        d = torch.from_numpy(np.ascontiguousarray(data)).cuda()
        o = torch.empty_like(d)
        m = vg.FilterDev(d.data_ptr(), len(data), stride, 0, o.data_ptr())
        torch.cuda.synchronize()
        want = o.cpu().numpy().reshape(len(data), -1)[:m]
        parts = [res[r][1][k] for r in range(WORLD)]
        assert all(p[0] == name for p in parts)
        got = np.concatenate([p[2] for p in parts])
        assert sum(p[1] for p in parts) == m, name
        assert got.tobytes() == want.tobytes(), name
        if name in ("plain", "chunked, one key"):  # a cloud that fills its box: every rank has a real share
            assert min(p[1] for p in parts) > m // (2 * WORLD), name
    whole = voxelgrid.New((0.05, 0.05, 0.05)).Filter(_cases()[0][1][:50_000])
    parts = [res[r][1][-1] for r in range(WORLD)]
    assert all(p[0] == "host" for p in parts) and sum(p[1] for p in parts) == whole.Points
    assert np.concatenate([p[2] for p in parts]).tobytes() == whole.Data.tobytes()
