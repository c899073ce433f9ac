import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
order = sys.argv[1]
import torch
if order == "torch_first":
    torch.cuda.set_device(0); torch.zeros(1, device="cuda")
from pcgol_amd import kdtree, synth
t = kdtree.New(synth.uniform_cloud(1000, 1.0, 1))
print("tree ok")
import torch.distributed as dist
print("avail", torch.cuda.is_available(), torch.cuda.device_count())
try:
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29577", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    x = torch.ones(3, device="cuda"); dist.all_reduce(x); print("nccl ok", x)
    dist.destroy_process_group()
except Exception as e:
    print("FAIL", type(e).__name__, str(e)[:100])
