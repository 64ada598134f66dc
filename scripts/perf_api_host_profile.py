"""Host profile (cProfile, by internal time) of compute_Sv and of compute_Sv + compute_MVBS on device-resident EK60
echodata, 4 x 500 000 x 2000 -- development aid: ~0.6 ms of Python per compute_Sv call, ~0.6 ms around the fused kernel."""
import cProfile, logging, pstats, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import echopype_amd as ep
C, P, S = 4, 500000, 2000
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
ed = ep.echodata.from_ek60_arrays(d).to_device()
for _ in range(3):
    ds = ep.calibrate.compute_Sv(ed); torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(20):
    ds = ep.calibrate.compute_Sv(ed)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
mv = lambda ds: ep.commongrid.compute_MVBS(ds, range_bin="1m", ping_time_bin="20s")
ds = ep.calibrate.compute_Sv(ed); m = mv(ds); torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(10):
    ds = ep.calibrate.compute_Sv(ed); m = mv(ds)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(25)
