"""Host profile of the reference's calls on device-resident EK60 echodata with the Sv deferred by compute_Sv
(4 x 500 000 x 2000): compute_Sv -> compute_MVBS, and compute_Sv -> remove_background_noise -> compute_MVBS -- development aid."""
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
ed = ep.echodata.from_ek60_arrays(d).to_device()


def timed(f, *a):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(*a); torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3, r


def prof(name, f, *a):
    pr = cProfile.Profile(); pr.enable(); r = f(*a); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter(); pr.disable()
    print(f"==== {name}: the final synchronize waited {1e3*(t2-t1):.2f} ms")
    pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
    return r


mvbs = lambda ds: ep.commongrid.compute_MVBS(ds, range_bin="1m", ping_time_bin="20s")  # noqa: E731
rbn = lambda ds: ep.clean.remove_background_noise(ds, ping_num=20, range_sample_num=50)  # noqa: E731
for rep in range(3):
    t_sv, ds = timed(ep.calibrate.compute_Sv, ed)
    t_mv, mv = timed(mvbs, ds)
    del ds, mv
    t_sv2, ds = timed(ep.calibrate.compute_Sv, ed)
    t_rbn, _ = timed(rbn, ds)
    c = ds.copy(); c["Sv"] = ds["Sv_corrected"]
    t_mv2, mv = timed(mvbs, c)
    del ds, mv, c
    print(f"compute_Sv {t_sv:.2f}  compute_MVBS(deferred) {t_mv:.2f} | compute_Sv {t_sv2:.2f}  remove_background_noise(deferred) {t_rbn:.2f}  compute_MVBS(Sv_corrected) {t_mv2:.2f} ms")
ds = ep.calibrate.compute_Sv(ed)
prof("compute_MVBS on a deferred Sv", mvbs, ds)
del ds
ds = ep.calibrate.compute_Sv(ed)
prof("remove_background_noise on a deferred Sv", rbn, ds)
