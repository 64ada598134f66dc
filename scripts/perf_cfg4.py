"""BASELINE configs[3] in full on one GPU: EK80 broadband, 2 ch x 200 000 pings x 8192 samples x 4 sectors,
complex samples held as float32 planes (105 GB), through the drop-in API (compute_Sv with pulse compression, then
compute_MVBS 20 s x 1 m) and at kernel level -- development aid / profiles."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import echopype_amd as ep  # noqa: E402
from echopype_amd import ops  # noqa: E402

C, P, S, B = (int(x) for x in (sys.argv[1:5] if len(sys.argv) > 4 else (2, 200000, 8192, 4)))
plane_dt = torch.float32 if (len(sys.argv) <= 5 or sys.argv[5] == "f32") else torch.float64
d = ep.synth.ek80_numpy(C, 4, 64, B)             # parameters only; the sample planes are generated on the device
g = torch.Generator(device="cuda"); g.manual_seed(20260504)
re = torch.empty((C, P, S, B), dtype=plane_dt, device="cuda")
im = torch.empty((C, P, S, B), dtype=plane_dt, device="cuda")
step = max(1, P // 50)
for p0 in range(0, P, step):                     # in slabs: no second copy of the 105 GB
    n = min(step, P - p0)
    re[:, p0:p0 + n] = (torch.randn((C, n, S, B), generator=g, device="cuda", dtype=torch.float32) * 1e-3).to(plane_dt)
    im[:, p0:p0 + n] = (torch.randn((C, n, S, B), generator=g, device="cuda", dtype=torch.float32) * 1e-3).to(plane_dt)
nan_pings = torch.rand(P, generator=g, device="cuda") < 0.10
tail = int(round(0.05 * S))
re[:, nan_pings, S - tail:] = float("nan")
im[:, nan_pings, S - tail:] = float("nan")
p = np.arange(P)
d.update(backscatter_r=ep.DeviceArray(re), backscatter_i=ep.DeviceArray(im),
         sample_interval=np.full((C, P), 8e-6), sound_speed=np.tile(1500.0 + 0.5 * np.sin(2 * np.pi * p / 1e5), (C, 1)),
         ping_time=np.datetime64("2026-05-01T00:00:00", "ns") + (p * 1_000_000_000).astype("timedelta64[ns]"))
ed = ep.echodata.from_ek80_arrays(d, ep.synth.ek80_filters())
n_out = C * P * S
print(f"EK80 BB {C} x {P} x {S} x {B}, planes {plane_dt}: {re.numel() * re.element_size() * 2 / 1e9:.1f} GB in", flush=True)


def run(dtype):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ds = ep.calibrate.compute_Sv(ed, waveform_mode="BB", encode_mode="complex", dtype=dtype)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    mv = ep.commongrid.compute_MVBS(ds, range_bin="1m", ping_time_bin="20s")
    torch.cuda.synchronize(); t2 = time.perf_counter()
    return ds, mv, t1 - t0, t2 - t1


for resident in (False, True):
  if resident:
    ed.to_device()  # the per-ping parameters into HBM once (EchoData.to_device)
  print("per-ping parameters", "in HBM" if resident else "on the host", flush=True)
  for dtype in ("float64", "float32"):
    run(dtype)
    ts = [run(dtype)[2:] for _ in range(3)]
    a, b = np.median([t[0] for t in ts]), np.median([t[1] for t in ts])
    print(f"API {dtype}: compute_Sv {a*1e3:8.1f} ms ({n_out/a/1e9:6.1f} Gsamp/s)   compute_MVBS {b*1e3:7.1f} ms   "
          f"both {n_out/(a+b)/1e9:6.1f} Gsamp/s", flush=True)
