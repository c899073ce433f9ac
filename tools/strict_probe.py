"""GPU probe: parallel strict sums (set_strict 1) against the one-wave chain (set_strict 2) on C4,
iteration by iteration, then timings.  Usage: python tools/strict_probe.py [n]"""
import os
import sys
import time

import numpy as np

os.environ["PCGX_STRICT_CLOCKS"] = "1"  # the us columns below; read when a session is created, dropped for the timings
sys.path.insert(0, ".")
from pcgol_amd import _lib as L, icp, kdtree, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
c = synth.c4_icp(n=n, width=10.0 * (n / 1e6) ** (1 / 3))
t = kdtree.New(c["base"])


def session(mode):
    s = icp.IcpSession(t, c["target"], c["max_dist"], c["min_pairs"], c["weight"], c["threshold"], c["max_iteration"])
    s.set_strict(mode)
    return s


a, b = session(1), session(2)
bad = 0
for k in range(c["max_iteration"]):
    a.step()
    b.step()
    sa, sb = a.read_sums(), b.read_sums()
    st = a.strict_stats()
    ok = np.array_equal(sa.view(np.uint64), sb.view(np.uint64))
    bad += not ok
    print("iter %2d %s runs %d runfail %d resolved %d (no aux %d) serial leaves %d recfail %d | us/row: phaseA %.1f walk %.1f (resolve aux %.1f serial %.1f)" % (
        k, "OK " if ok else "MISMATCH", st[0], st[1], st[2], st[5], st[3], st[4], st[8] / 900.0, st[9] / 900.0, st[10] / 900.0, st[11] / 900.0), "slowest row walk %.1f us" % (st[46] / 100.0), "selfcheck bad: run %d prefix %d tile %d suffix %d; scan != serial composition %d, other result %d" % (tuple(st[12:16]) + (st[6], st[7])))
    print("      no window: %d tiles, %d runs tried, %d applied, %d leaves serial, %.1f us each (candidate hits %d of %d) | crossing: %d tiles, %d leaves serial, %.1f us each" % (
        st[16], st[17], st[18], st[22], st[19] / 100.0 / max(st[16], 1), st[25], st[24], st[20], st[23], st[21] / 100.0 / max(st[20], 1)))
    print("      walker, us/row: in failed runs %.1f (waiting for a helper %.1f, %d times; for the backward scan %.1f); fetching the runs' records %.1f, all runs %.1f" % (st[36] / 900.0, st[37] / 900.0, st[38], st[39] / 900.0, st[40] / 900.0, st[41] / 900.0))
    print("      later chunks' walkers waited %.1f us in all (the rows' last chunks %.1f us per row); walked alone %d" % (st[60] / 100.0, st[61] / 100.0 / 9, st[62]))
    print("      tiles handed to a helper: table not up %d, state outside the table %d, no such candidate %d" % (st[42], st[43], st[44]))
    if st[46] / 100.0 > 20.0:
        print("      walk per row, us:", " ".join("%.1f" % (st[27 + r] / 100.0) for r in range(9)))
    if any(st[48:58]):
        print("      states that missed their tile's table, by log2 of the distance to the guess (0: equal .. 8: >= 2^7, 9: other sign):", " ".join("%d:%d" % (b, st[48 + b]) for b in range(10) if st[48 + b]))
    if st[58]:
        print("      plain tiles the repair pass turned into jobs: %d" % st[58])
    if st[5]:
        print("      last tile without a slot that was recomputed: %d, guess %08x, state %08x" % (st[45], int(st[47]) >> 32, int(st[47]) & 0xffffffff))
    for q in range(0):
        d = st[16 + 16 * q: 32 + 16 * q]
        if d[0] or d[1]:
            print("      s_in %08x s(Qr) %08x y(acc) %08x key %x n %d c %d lo %d hi %d acc.c %d | again: ok %d z %08x | in %08x out %08x cons %d" % tuple(int(v) & 0xffffffff for v in d[:14]))
    if not ok and 0:
        print("   parallel", sa)
        print("   serial  ", sb)
ta, _, _ = a.result()
tb, _, _ = b.result()
print("final pose equal:", np.array_equal(ta, tb), "mismatching iterations:", bad)
del os.environ["PCGX_STRICT_CLOCKS"]
for mode in (1, 2, 0):
    s = session(mode)
    for rep in range(2):
        L.check(L.lib().pcgx_icp_session_reset(s._h, None))
        L.check(L.lib().pcgx_sync(None))
        t0 = time.perf_counter()
        for k in range(c["max_iteration"]):
            s.step()
        L.check(L.lib().pcgx_sync(None))
        dt = (time.perf_counter() - t0) / c["max_iteration"]
    print("strict %d: %.1f us per iteration, %.2f Gpoints/s" % (mode, dt * 1e6, n / dt / 1e9))
