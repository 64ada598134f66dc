import cProfile, logging, pstats, sys, time
import numpy as np
import torch
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
f = lambda: ep.compute_Sv_clean_MVBS(ed, 20, 50, range_bin="1m", ping_time_bin="20s")
r = f(); torch.cuda.synchronize()
for _ in range(2):
    del r; torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"host done after {1e3*(t1-t0):.2f} ms, kernels done after {1e3*(t2-t0):.2f} ms")
del r
pr = cProfile.Profile(); pr.enable(); r = f(); pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
