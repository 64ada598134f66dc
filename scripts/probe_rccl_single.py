"""World-size-1 RCCL probe (1-GPU box): the exact init / collective calls bench.py and echopype_amd.sharding
issue for N > 1 -- init_process_group("nccl", device_id=...), barrier, MAX all-reduce of the timing scalar,
SUM all-reduce of edge-bin (sum, count) partials -- so that a missing library or a bad argument shows up
before the driver's multi-GPU run."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
from echopype_amd import sharding  # noqa: E402

dist.barrier()
t = torch.tensor([3.25], dtype=torch.float64, device="cuda")
dist.all_reduce(t, op=dist.ReduceOp.MAX)
assert t.item() == 3.25
print("global_max", sharding.global_max(7.5))
s = torch.arange(12, dtype=torch.float64, device="cuda").reshape(2, 2, 3)
c = torch.ones((2, 2, 3), dtype=torch.int32, device="cuda")
dist.all_reduce(s)
dist.all_reduce(c)
torch.cuda.synchronize()
print("rccl ok", torch.cuda.get_device_name(0), dist.get_backend(), s.sum().item(), c.sum().item())
dist.barrier()
dist.destroy_process_group()
