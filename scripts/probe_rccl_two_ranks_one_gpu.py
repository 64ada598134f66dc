"""Can RCCL run TWO ranks on ONE GPU (1-GPU boxes)?  If it can, the edge exchange gets a real peer; if it refuses
("duplicate GPU"), world size 1 is all a 1-GPU box can execute of the NCCL branch -- development aid.
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29571 scripts/probe_rccl_two_ranks_one_gpu.py"""
import os
import sys

import torch
import torch.distributed as dist

torch.cuda.set_device(0)
try:
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    t = torch.full((4,), float(dist.get_rank() + 1), dtype=torch.float64, device="cuda")
    dist.all_reduce(t)
    torch.cuda.synchronize()
    print("rank", dist.get_rank(), "all_reduce over two ranks on one GPU:", t.tolist(), flush=True)
    dist.destroy_process_group()
except Exception as e:  # noqa: BLE001
    print("rank", os.environ.get("RANK"), "RCCL refused:", repr(e)[:300], flush=True)
    sys.exit(3)
