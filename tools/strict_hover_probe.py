"""Strict sums, worst case: a target whose gradient sums hover around zero from the first pair to the
last (base points + zero-mean noise, no transform).  Time per iteration and what the chain did."""
import os
import sys
import time

import numpy as np


sys.path.insert(0, ".")
from pcgol_amd import _lib as L, icp, kdtree, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
f32 = np.float32
rng = np.random.Generator(np.random.PCG64(77))
base = synth.uniform_cloud(n, 10.0 * (n / 1e6) ** (1 / 3), 2)
noise = ((rng.integers(0, 1 << 16, size=(n, 3)).astype(np.float32) / f32(1 << 16)) - f32(0.5)) * f32(0.02)
target = np.ascontiguousarray((base[rng.permutation(n)] + noise).astype(np.float32))
t = kdtree.New(base)
for mode in (1, 0):
    os.environ.pop("PCGX_STRICT_CLOCKS", None)  # (read when a session is created: the timed one runs without the clock reads)
    s = icp.IcpSession(t, target, 0.5, 6, np.full(6, 0.3, f32), np.full(6, -1.0, f32), 20)
    s.set_strict(mode)
    for _ in range(20):
        s.step()
    L.check(L.lib().pcgx_sync(None))
    if mode:
        s.strict_stats()
    L.check(L.lib().pcgx_icp_session_reset(s._h, None))
    t0 = time.perf_counter()
    for _ in range(20):
        s.step()
    L.check(L.lib().pcgx_sync(None))
    dt = (time.perf_counter() - t0) / 20
    extra = ""
    if mode:
        st = s.strict_stats()
        extra = " | per iteration: runs %.0f, runs not covered %.0f, tiles recomputed %.0f, leaves term by term %.0f" % (
            st[0] / 20, st[1] / 20, st[2] / 20, st[3] / 20)
        extra += " | us per sum: scans %.1f, walk %.1f (recomputing %.1f), slowest walk of the last launch %.1f" % (
            st[8] / 180 / 100.0, st[9] / 180 / 100.0, (st[10] + st[11]) / 180 / 100.0, st[46] / 100.0)
    print("strict %d: %.1f us per iteration%s" % (mode, dt * 1e6, extra))
    if mode:  # per iteration (a read-back after every step: the clocks differ from the timed loop's)
        s.close()
        os.environ["PCGX_STRICT_CLOCKS"] = "1"
        s = icp.IcpSession(t, target, 0.5, 6, np.full(6, 0.3, f32), np.full(6, -1.0, f32), 20)
        s.set_strict(mode)
        for k in range(20):
            s.step()
            st = s.strict_stats()
            print("  iteration %2d: tiles recomputed %3d, leaves term by term %4d, slowest sum's walk %.0f us | no window: %d tiles, %d runs tried, %d applied, %d leaves serial, %.1f us each (table hits %d exact + %d between, of %d) | crossing: %d tiles, %d leaves serial, %.1f us each | no slot %d" % (
                k, st[2], st[3], st[46] / 100.0, st[16], st[17], st[18], st[22], st[19] / 100.0 / max(st[16], 1), st[25], st[26], st[24], st[20], st[23], st[21] / 100.0 / max(st[20], 1), st[5]))
            if sum(st[48:64]):
                print("      table misses by log2(distance of the state from the guess): %s, other sign %d" % (list(int(v) for v in st[48:57]), st[57]))
    s.close()
