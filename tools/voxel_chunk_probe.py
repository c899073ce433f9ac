"""Voxel filter timing, non-chunked vs WithChunkSize (device resident, C3 shape)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pcgol_amd import synth, voxelgrid, _lib as L
c3 = synth.c3_voxel()
n = len(c3["points"])
L.check(L.lib().pcgx_init(0))
dp = torch.from_numpy(c3["points"]).cuda()
dout = torch.empty_like(dp)
st = torch.cuda.current_stream().cuda_stream
for name, opts in (("non-chunked", ()), ("chunk 64^3", (voxelgrid.WithChunkSize([64, 64, 64]),)),
                   ("chunk 16^3", (voxelgrid.WithChunkSize([16, 16, 16]),))):
    vg = voxelgrid.New(c3["leaf"], *opts)
    for _ in range(2):
        m = vg.FilterDev(dp.data_ptr(), n, 12, 0, dout.data_ptr(), st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        m = vg.FilterDev(dp.data_ptr(), n, 12, 0, dout.data_ptr(), st)
    torch.cuda.synchronize()
    print("%s: %.3f ms/call, M = %d" % (name, (time.perf_counter() - t0) / 5 * 1e3, m))
