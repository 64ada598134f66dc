import os, sys, time, torch, torch.distributed as dist, torch.multiprocessing as mp
def w(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    t = torch.zeros(2, dtype=torch.int64)
    for _ in range(50): dist.all_reduce(t)
    dist.barrier(); t0 = time.perf_counter()
    n = 500
    for _ in range(n): dist.all_reduce(t, op=dist.ReduceOp.MIN)
    dt = (time.perf_counter() - t0) / n * 1e6
    if rank == 0: print(f"world {world}: {dt:.0f} us per 16-byte all-reduce", flush=True)
    dist.destroy_process_group()
if __name__ == "__main__":
    for world in (2, 4, 8):
        mp.spawn(w, args=(world, 29600 + world), nprocs=world, join=True)
