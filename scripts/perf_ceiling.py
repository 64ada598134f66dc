"""Empirical HBM ceilings for the traffic mixes of the hot path (development aid)."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from echopype_amd import ops
n = 4 * 100000 * 2000
x = torch.rand(n, device="cuda", dtype=torch.float32)
y = torch.empty(n, device="cuda", dtype=torch.float64)
z = torch.empty(n, device="cuda", dtype=torch.float64)
y32 = torch.empty(n, device="cuda", dtype=torch.float32)
t = ops.Timer()
def timeit(name, fn, nbytes):
    fn(); torch.cuda.synchronize(); ms = []
    for _ in range(7):
        t.start(); fn(); t.stop(); ms.append(t.elapsed_ms())
    m = float(np.median(ms)); print(f"{name:42s} {m:8.3f} ms {nbytes/m/1e9:6.2f} TB/s", flush=True)
timeit("torch f32->f64 cast (4B in, 8B out)", lambda: y.copy_(x), n * 12)
timeit("torch f64 copy (8B in, 8B out)", lambda: z.copy_(y), n * 16)
timeit("torch f32 copy (4B in, 4B out)", lambda: y32.copy_(x), n * 8)
timeit("torch f64 fill (8B out)", lambda: z.fill_(1.0), n * 8)
timeit("torch f64 sum (8B in)", lambda: y.sum(), n * 8)
timeit("torch f32 sum (4B in)", lambda: x.sum(), n * 4)
