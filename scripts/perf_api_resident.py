"""The Dataset API on DEVICE-resident echodata at the headline volume (EK60 4 x 500 000 x 2000): what the host-side
parameter assembly and Dataset bookkeeping cost on top of the kernels (development aid)."""
import sys, time, logging
import numpy as np
import torch
sys.path.insert(0, ".")
import echopype_amd as ep
C, P, S = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (4, 500000, 2000)))
dd = ep.synth.ek60_device(C, P, S)
d = ep.synth.ek60_numpy(C, 4, 8)  # host-side parameter layout; per-ping vectors re-made at full length below
p = np.arange(P)
for k, v in list(d.items()):
    if isinstance(v, np.ndarray) and v.ndim == 2 and v.shape == (C, 4):
        d[k] = np.repeat(v[:, :1], P, axis=1)
d["sound_speed_indicative"] = np.tile(1500.0 + 0.5 * np.sin(2 * np.pi * p / 1e5), (C, 1)) if np.ndim(d["sound_speed_indicative"]) == 2 else d["sound_speed_indicative"]
d["backscatter_r"] = ep.DeviceArray(dd["backscatter_r"])
d["ping_time"] = np.datetime64("2026-05-01T00:00:00", "ns") + (p * 1_000_000_000).astype("timedelta64[ns]")
ed = ep.echodata.from_ek60_arrays(d).to_device()
n = C * P * S
logging.disable(logging.WARNING)
def t(f):
    r = f(); torch.cuda.synchronize(); ts = []
    for _ in range(4):
        del r  # one result resident at a time: the caching allocator hands the same blocks back
        torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return float(np.median(ts)), r
# (compute_Sv defers the Sv array to its first reader: the two calls are timed together, on a fresh dataset each time;
#  then compute_MVBS alone on a dataset whose Sv has been written)
def two_calls():
    ds_ = ep.calibrate.compute_Sv(ed)
    return ds_, ep.commongrid.compute_MVBS(ds_, range_bin="1m", ping_time_bin="20s")
a, (ds, mv) = t(two_calls)
b, mv = t(lambda: ep.commongrid.compute_MVBS(ds, range_bin="1m", ping_time_bin="20s"))
print(f"compute_Sv + compute_MVBS (Sv deferred) {a*1e3:7.1f} ms = {n/a/1e9:6.1f} Gsamp/s      compute_MVBS on the written Sv {b*1e3:7.1f} ms = {n/b/1e9:6.1f} Gsamp/s")
del ds, mv
c, r = t(lambda: ep.compute_Sv_MVBS(ed, range_bin="1m", ping_time_bin="20s"))
print(f"compute_Sv_MVBS (one pass) {c*1e3:7.1f} ms = {n/c/1e9:6.1f} Gsamp/s")
del r
e, r = t(lambda: ep.compute_Sv_clean_MVBS(ed, 20, 50, range_bin="1m", ping_time_bin="20s"))
print(f"compute_Sv_clean_MVBS (two passes) {e*1e3:7.1f} ms = {n/e/1e9:6.1f} Gsamp/s")
# ---- the steps either side, on the Sv dataset of the first 100 000 pings
del r
Pm = min(P, 100000)
d2 = dict(d); d2["backscatter_r"] = ep.DeviceArray(dd["backscatter_r"][:, :Pm].contiguous())
for k, v in list(d2.items()):
    if isinstance(v, np.ndarray) and v.ndim == 2 and v.shape == (C, P):
        d2[k] = v[:, :Pm]
d2["ping_time"] = d["ping_time"][:Pm]
ds = ep.calibrate.compute_Sv(ep.echodata.from_ek60_arrays(d2))
nm = C * Pm * S
for name, f in (
    ("remove_background_noise(20, 50)", lambda: ep.clean.remove_background_noise(ds, 20, 50)),
    ("estimate_background_noise", lambda: ep.clean.estimate_background_noise(ds, 20, 50)),
    ("mask_impulse_noise (index)", lambda: ep.clean.mask_impulse_noise(ds, range_var="echo_range", use_index_binning=True)),
    ("mask_transient_noise (index)", lambda: ep.clean.mask_transient_noise(ds, range_var="echo_range", use_index_binning=True, exclude_above="20.0m")),
    ("mask_transient_noise (value windows, default)", lambda: ep.clean.mask_transient_noise(ds, range_var="echo_range", exclude_above="20.0m")),
    ("mask_attenuated_signal", lambda: ep.clean.mask_attenuated_signal(ds, range_var="echo_range", upper_limit_sl="150.0m", lower_limit_sl="250.0m")),
    ("compute_MVBS_index_binning", lambda: ep.commongrid.compute_MVBS_index_binning(ds, 100, 100)),
    ("add_depth(depth_offset, tilt)", lambda: ep.consolidate.add_depth(ds, depth_offset=5.0, tilt=10.0)),
):
    try:
        tt, _ = t(f)
        print(f"{name:48s} {tt*1e3:8.1f} ms = {nm/tt/1e9:7.1f} Gsamp/s", flush=True)
    except Exception as ex:  # noqa: BLE001
        print(name, "failed:", type(ex).__name__, ex, flush=True)
