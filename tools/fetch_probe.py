"""What rocprofv3's FETCH_SIZE / WRITE_SIZE report on this GPU for access patterns with a known
byte count (run under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE`, see profiles/collect.sh):
  copy    : 1 GiB streamed device to device (wide coalesced 16-byte loads)
  gather16: 16M random 16-byte rows out of a 1 GiB table (torch.index_select on a [N, 4] float32 tensor)
  gather4 : 64M random 4-byte elements out of the same table
The summary (profiles/summarize.py) turns the counters into the factor that FETCH_SIZE has to be
multiplied with to give bytes, per pattern."""
import torch

torch.manual_seed(1)
n_rows = 1 << 26          # x 16 B = 1 GiB
table = torch.rand(n_rows, 4, device="cuda")
dst = torch.empty_like(table)
idx16 = torch.randint(0, n_rows, (1 << 24,), device="cuda")
idx4 = torch.randint(0, n_rows * 4, (1 << 26,), device="cuda")
flat = table.view(-1)
torch.cuda.synchronize()
for _ in range(3):
    dst.copy_(table)
    torch.cuda.synchronize()
    g16 = torch.index_select(table, 0, idx16)
    torch.cuda.synchronize()
    g4 = torch.index_select(flat, 0, idx4)
    torch.cuda.synchronize()
print("copy bytes", table.numel() * 4, "gather16 rows", idx16.numel(), "gather4 elements", idx4.numel())
