"""Where the Dataset API spends its time on device-resident echodata (development aid): compute_Sv_MVBS on EK60
4 x 500 000 x 2000 with the samples AND the per-ping parameters in HBM (EchoData.to_device), cProfile of one call."""
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
for resident in (False, True):
    ed = ep.echodata.from_ek60_arrays(d)
    if resident:
        ed.to_device()
    f = lambda: ep.compute_Sv_MVBS(ed, range_bin="1m", ping_time_bin="20s")
    r = f(); torch.cuda.synchronize(); ts = []
    for _ in range(5):
        del r; torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print(f"parameters {'in HBM' if resident else 'on the host'}: compute_Sv_MVBS {np.median(ts)*1e3:7.2f} ms = {n/np.median(ts)/1e9:6.1f} Gsamp/s")
    del r
pr = cProfile.Profile(); pr.enable(); r = f(); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
