"""The kNN search kernel's time against the number of queries (1M-point tree of C2): is it a staircase -- whole rounds
of resident workgroups -- or a line?   python tools/knn_nq_sweep.py"""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
from pcgol_amd import kdtree, synth, _lib as L
pts = synth.uniform_cloud(1_000_000, 10.0, 2)
t = kdtree.New(pts)
qall = synth.uniform_cloud(1_600_000, 10.0, 3)
dq = torch.from_numpy(qall).cuda()
ids = torch.empty(len(qall), dtype=torch.int32, device="cuda"); dsq = torch.empty(len(qall), dtype=torch.float32, device="cuda")
for n in (300_000, 327_680, 400_000, 500_000, 600_000, 655_360, 700_000, 800_000, 900_000, 983_040, 1_000_000, 1_100_000, 1_200_000, 1_310_720, 1_400_000, 1_600_000):
    for _ in range(3):
        t.NearestBatchDev(dq.data_ptr(), n, 10.0, ids.data_ptr(), dsq.data_ptr(), presort=True)
    torch.cuda.synchronize()
    L.prof_enable(1); L.prof_reset()
    for _ in range(10):
        t.NearestBatchDev(dq.data_ptr(), n, 10.0, ids.data_ptr(), dsq.data_ptr(), presort=True)
    torch.cuda.synchronize()
    ms, cnt = L.prof_read(L.PROF_KNN_GRID)
    L.prof_enable(0)
    us = ms / max(cnt, 1) * 1e3
    print(f"{n:8d} queries: search kernel {us:6.1f} us  = {us * 1e3 / n:6.2f} ns per 1000 queries ... {n / 256 / 1280:5.2f} rounds of 1280 workgroups")
