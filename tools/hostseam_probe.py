"""Where a host-pointer VoxelGrid C3 call spends its time: upload, filter, download, each by itself."""
import ctypes as C, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pcgol_amd import synth, _lib as L
c3 = synth.c3_voxel()
pts = c3["points"]; n = len(pts)
L.check(L.lib().pcgx_init(0))
dp = torch.empty(n * 3, dtype=torch.float32, device="cuda"); dout = torch.empty_like(dp)
out = np.zeros_like(pts); m = C.c_int64()
leaf = (C.c_float * 3)(*c3["leaf"]); chunk = (C.c_int32 * 3)(0, 0, 0)
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
up = t(lambda: L.check(L.lib().pcgx_dev_upload(C.c_void_p(dp.data_ptr()), L.ptr(pts), n * 12)))
fl = t(lambda: L.check(L.lib().pcgx_voxel_filter_dev(C.c_void_p(dp.data_ptr()), n, 12, 0, leaf, chunk, C.c_void_p(dout.data_ptr()), C.byref(m), None)))
dn = t(lambda: L.check(L.lib().pcgx_dev_download(L.ptr(out), C.c_void_p(dout.data_ptr()), m.value * 12)))
al = t(lambda: L.check(L.lib().pcgx_voxel_filter(L.ptr(pts), n, 12, 0, leaf, chunk, L.ptr(out), C.byref(m))))
print("upload %.3f ms (%.1f GB/s)  filter %.3f ms  download %.3f ms (%.1f GB/s)  sum %.3f  | pcgx_voxel_filter %.3f ms" % (
    up, n * 12 / up / 1e6, fl, dn, m.value * 12 / dn / 1e6, up + fl + dn, al))
# pinned source for comparison
pp = torch.from_numpy(pts).pin_memory()
upp = t(lambda: L.check(L.lib().pcgx_dev_upload(C.c_void_p(dp.data_ptr()), C.c_void_p(pp.data_ptr()), n * 12)))
print("upload from pinned memory %.3f ms (%.1f GB/s)" % (upp, n * 12 / upp / 1e6))
