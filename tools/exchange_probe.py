"""Per-step time of the N > 1 code path (partials -> all-reduce -> update) on ONE GPU through a
1-rank RCCL group, next to the fused single-GPU step.  python tools/exchange_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.distributed as dist
from pcgol_amd import _lib as L, kdtree, synth
from pcgol_amd.distributed import ShardedIcp
torch.cuda.set_device(0)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
L.check(L.lib().pcgx_init(0))
c = synth.c4_icp()
t = kdtree.New(c["base"])
for name, force in (("fused step (N = 1)", False), ("partials -> all_reduce -> update (1-rank RCCL)", True)):
    s = ShardedIcp(t, c["target"], c["max_dist"], c["min_pairs"], c["weight"], c["threshold"], c["max_iteration"],
                   force_exchange=force)
    for _ in range(40): s.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(100):
        if k % 20 == 0: s.reset()
        s.step()
    torch.cuda.synchronize()
    print("%s: %.4f ms/step" % (name, (time.perf_counter() - t0) / 100 * 1e3))
    s.close()
dist.destroy_process_group()
