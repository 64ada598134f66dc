"""Host profile (cProfile, by internal time) of ``sharding.compute_Sv_MVBS`` on a one-rank RCCL group: the host cost of
the N > 1 route per tile -- control messages, exchange plan, collectives as identities -- next to the two reference
calls on the same tiles.  Development aid (round 5): python scripts/perf_shard_host_profile.py [tile pings]"""
import argparse, cProfile, logging, os, pstats, sys, time
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, ".")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
import bench
tile_p = int(sys.argv[1]) if len(sys.argv) > 1 else 250_000
for sharded in (False, True):
    args = argparse.Namespace(dtype="float64", steps=1, warmup=0, passes=None, ss_every=1, backend="nccl", sharded_at_1=sharded,
                              single_device=False)
    ctx = bench.Ctx(args, 1, 0)
    job = bench.Cfg5(ctx, 4, 2 * tile_p, 4096, tile_pings=tile_p)
    one_pass, finish, state, eds = job.api_layout(10_000_000_000)
    logging.disable(logging.WARNING)
    for _ in range(3):
        one_pass(None)
    finish(); torch.cuda.synchronize()
    t = bench.ops.Timer() if hasattr(bench, "ops") else None
    class T:  # (a timer stand-in: host_s is only accumulated on timed passes)
        def start(self): pass
        def stop(self): pass
    n = 10
    t0 = time.perf_counter()
    pr = cProfile.Profile(); pr.enable()
    for _ in range(n):
        one_pass(T())
    pr.disable()
    host = time.perf_counter() - t0
    finish(); torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    print(f"== sharded={sharded}: {state['host_s'] / state['calls'] * 1e3:.3f} ms of host time per tile's calls "
          f"({host / n / 2 * 1e3:.3f} ms per tile incl. the late reads), {wall / n / 2 * 1e3:.2f} ms wall per tile", flush=True)
    pstats.Stats(pr).sort_stats("tottime").print_stats(28)
    del one_pass, finish, eds, job
    ctx.free()
dist.destroy_process_group()
