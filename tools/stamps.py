"""Per-workgroup wall clock stamps of an instrumented kernel (csrc/wg_stamps.h): builds nothing itself -- run it with a
library built with -DPCGX_STAMPS:
    bash tools/mk_variant.sh stamps "-DPCGX_STAMPS" && PCGX_LIB=experiments/ab/libpcgx_stamps.so python tools/stamps.py vb_bucket
Runs the C3 filter a few times, reads the last call's stamps, prints every phase's time per workgroup (mean / median /
p95), the workgroups' lives and how many were at work at a time."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pcgol_amd import synth, voxelgrid, _lib as L

KERNELS = {  # name -> (stamps per workgroup, rows to read, phase names (stamp k - stamp k-1), driver)
    "vb_bucket": (8, 65536, ["bounds", "points asked for, counts cleared, barrier", "ranks (ballots)", "cells scanned",
                             "points to their places", "cell phase, stores issued"], "voxel"),
}


def run_voxel():
    n = 10_000_000
    c3 = synth.c3_voxel(n)
    dp = torch.from_numpy(c3["points"]).cuda()
    dout = torch.empty_like(dp)
    vg = voxelgrid.New(c3["leaf"])
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(6):
        vg.FilterDev(dp.data_ptr(), n, 12, 0, dout.data_ptr(), st)
    torch.cuda.synchronize()


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "vb_bucket"
    per, rows, phases, driver = KERNELS[name]
    L.check(L.lib().pcgx_init(0))
    raw = ctypes.CDLL(L.lib()._name)
    fn = getattr(raw, "pcgx_debug_stamps_" + name, None)
    if fn is None:
        raise SystemExit("this library has no stamps for %s: build it with -DPCGX_STAMPS (tools/mk_variant.sh) and set PCGX_LIB" % name)
    {"voxel": run_voxel}[driver]()
    out = (ctypes.c_ulonglong * (rows * per))()
    if fn(out, rows) != 0:
        raise SystemExit("reading the stamps failed")
    v = np.frombuffer(out, dtype=np.uint64).astype(np.float64).reshape(rows, per) / 100.0  # us
    last = len(phases)
    v = v[(v[:, 0] > 0) & (v[:, last] > 0)]
    print("%s: %d workgroup-units stamped" % (name, len(v)))
    for k, nm in enumerate(phases, 1):
        d = v[:, k] - v[:, k - 1]
        print("  %-44s mean %6.2f us  median %6.2f  p95 %6.2f" % (nm, d.mean(), np.median(d), np.percentile(d, 95)))
    life = v[:, last] - v[:, 0]
    span = v[:, last].max() - v[:, 0].min()
    print("  a unit: mean %.2f median %.2f us; first in -> last out %.1f us; at work at a time: %.0f" % (life.mean(), np.median(life), span, life.sum() / span))


if __name__ == "__main__":
    main()
