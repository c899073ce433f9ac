"""n session-resident C4 Fits from n host threads, on the library's stream or on a stream each:
    python tools/conc_resident_probe.py 4 [own]"""
import os, sys, time, threading
sys.path.insert(0, os.getcwd())
import numpy as np
import torch
from pcgol_amd import synth, icp, kdtree, _lib as L
c4 = synth.c4_icp()
tree = kdtree.New(c4["base"])
def mk():
    return icp.IcpSession(tree, c4["target"], c4["max_dist"], c4["min_pairs"], c4["weight"], c4["threshold"], c4["max_iteration"])
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
ss = [mk() for _ in range(n)]
own = len(sys.argv) > 2 and sys.argv[2] == 'own'
streams = [torch.cuda.Stream().cuda_stream if own else 0 for _ in range(n)]
def fit(s, st=0):
    L.check(L.lib().pcgx_icp_session_reset(s._h, L.ptr(st) if st else None))
    for _ in range(20):
        s.step(st)
    s.result(st)
for s, st in zip(ss, streams): fit(s, st)
t0 = time.perf_counter(); fit(ss[0], streams[0]); one = time.perf_counter() - t0
for rep in range(5):
    th = [threading.Thread(target=fit, args=(s, st)) for s, st in zip(ss, streams)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    dt = time.perf_counter() - t0
    print(("own streams, " if own else "the library's stream, ") + "resident: one %.3f ms; %d at once %.3f ms (%.2f x one)" % (one*1e3, n, dt*1e3, dt/one))
