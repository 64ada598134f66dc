"""compute_Sv then compute_MVBS (the reference's two calls) on device-resident EK60 echodata, 4 x 500 000 x 2000:
wall time of each, the kernels' share (HIP events around the calls' launches are not available from outside, so
the host profile of one call each is printed instead) -- development aid."""
import cProfile, logging, pstats, sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
import echopype_amd as ep
C, P, S = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (4, 500000, 2000)))
dd = ep.synth.ek60_device(C, P, S)
d = ep.synth.ek60_numpy(C, 4, 8)
p = np.arange(P)
for k, v in list(d.items()):
    if isinstance(v, np.ndarray) and v.ndim == 2 and v.shape == (C, 4):
        d[k] = np.repeat(v[:, :1], P, axis=1)
d["sound_speed_indicative"] = np.tile(1500.0 + 0.5 * np.sin(2 * np.pi * p / 1e5), (C, 1))
d["backscatter_r"] = ep.DeviceArray(dd["backscatter_r"])
d["ping_time"] = ep.synth.T0 + (p * 1_000_000_000).astype("timedelta64[ns]")
logging.disable(logging.WARNING)
n = C * P * S
ed = ep.echodata.from_ek60_arrays(d).to_device()
def med(f, keep=False):
    r = f(); torch.cuda.synchronize(); ts = []
    for _ in range(5):
        del r; torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return float(np.median(ts)), r
a, ds = med(lambda: ep.calibrate.compute_Sv(ed))
b, mv = med(lambda: ep.commongrid.compute_MVBS(ds, range_bin="1m", ping_time_bin="20s"))
print(f"compute_Sv {a*1e3:.2f} ms   compute_MVBS {b*1e3:.2f} ms")
del mv
for name, f in (("compute_Sv", lambda: ep.calibrate.compute_Sv(ed)), ("compute_MVBS", lambda: ep.commongrid.compute_MVBS(ds, range_bin="1m", ping_time_bin="20s"))):
    if name == "compute_Sv":
        del ds
    pr = cProfile.Profile(); pr.enable(); r = f(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter(); pr.disable()
    print(f"==== {name}: the final synchronize waited {1e3*(t2-t1):.2f} ms (kernels still running when the host was done)")
    pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
    if name == "compute_Sv":
        ds = r
