"""Times the voxel filter (device resident, C3 shape) -- helper for rocprofv3 runs."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pcgol_amd import synth, voxelgrid, _lib as L
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
c3 = synth.c3_voxel(n)
L.check(L.lib().pcgx_init(0))
dp = torch.from_numpy(c3["points"]).cuda()
dout = torch.empty_like(dp)
vg = voxelgrid.New(c3["leaf"])
st = torch.cuda.current_stream().cuda_stream
for _ in range(2):
    m = vg.FilterDev(dp.data_ptr(), n, 12, 0, dout.data_ptr(), st)
torch.cuda.synchronize()
# VOXEL_PROBE_BETWEEN: 1 = the stream drained between calls, 2 = and a 1.5 GB read sweep (the caches full of clean lines
# of something else), 3 = and a 1.5 GB write sweep (full of dirty ones) -- for the timeline, the ms/call then means nothing
between = int(os.environ.get("VOXEL_PROBE_BETWEEN", "0"))
big = torch.zeros(375_000_000, dtype=torch.float32, device="cuda") if between >= 2 else None
t0 = time.perf_counter()
for _ in range(40):
    m = vg.FilterDev(dp.data_ptr(), n, 12, 0, dout.data_ptr(), st)
    if between:
        torch.cuda.synchronize()
        if between == 2:
            big.sum()
        if between == 3:
            big.add_(1.0)
        torch.cuda.synchronize()
torch.cuda.synchronize()
print("voxel ms/call", (time.perf_counter() - t0) / 40 * 1e3, "M", m)
