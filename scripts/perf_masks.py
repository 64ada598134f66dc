"""Timing probe of the noise-mask kernels (HIP events) -- development aid, not the bench."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from echopype_amd import ops, synth

C, P, S = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (4, 100000, 2000)))
d = synth.ek60_device(C, P, S)
cf = ops.power_coef_ek(d["sample_interval"], d["transmit_duration_nominal"], d["transmit_power"],
    d["sound_speed_indicative"], d["absorption_indicative"], d["gain_correction"], d["sa_correction"],
    d["equivalent_beam_angle"], d["frequency_nominal"], d["transmit_duration_nominal"][:, 0].contiguous(),
    pulse_length=d["pulse_length"], gain_is_table=True, sa_is_table=True)
t = ops.Timer()
def timeit(name, fn, n, bytes_per_sample, reps=3):
    fn(); torch.cuda.synchronize()
    ms = []
    for _ in range(reps):
        t.start(); fn(); t.stop(); ms.append(t.elapsed_ms())
    m = float(np.median(ms))
    print(f"{name:46s} {m:10.3f} ms  {n/m/1e6:9.3f} Gsamp/s  {n*bytes_per_sample/m/1e9:7.2f} TB/s (algorithmic)", flush=True)
    return m
for dt, b in ((torch.float64, 8), (torch.float32, 4)):
    sv, rng = ops.sv_power(d["backscatter_r"], cf, dtype=dt)
    n = sv.numel()
    step = float(ops.range_step_mean(rng)[0])
    n5, n10 = int(np.ceil(5 / step)), int(np.ceil(10 / step))
    print(f"-- {dt}: step {step:.4f} m, 5 m = {n5} samples, 10 m = {n10} samples", flush=True)
    timeit(f"range_step_mean", lambda: ops.range_step_mean(rng), n, b)
    timeit(f"range_bin_smooth index (5 m)", lambda: ops.range_bin_smooth(sv, nper=n5), n, 2 * b)
    lo, hi = ops.nanminmax(rng)
    nb = len(np.arange(lo, hi + 5.0, 5.0)) - 1
    up = ops.range_bin_smooth(sv, range=rng, r0=lo, bin=5.0, nbins=nb)
    timeit(f"range_bin_smooth value (5 m, {nb} bins)", lambda: ops.range_bin_smooth(sv, range=rng, r0=lo, bin=5.0, nbins=nb), n, 3 * b)
    timeit(f"impulse_mask (n=2)", lambda: ops.impulse_mask(up, 2, 10.0), n, b + 1)
    del up
    timeit(f"pool_sv nanmean 51 x {2*n10+1} -> mask", lambda: ops.pool_sv(sv, 100, 25, n10, threshold=12.0, want_pooled=False), n, b + 1)
    timeit(f"pool_sv nanmean 5 x 11 -> mask", lambda: ops.pool_sv(sv, 100, 2, 5, threshold=12.0, want_pooled=False), n, b + 1)
    W = int(100 / step)
    timeit(f"attenuated_mask (n=15, layer {W} samples)", lambda: ops.attenuated_mask(sv, rng, 150.0, 250.0, 15, -6.0), n, 2 * b + 1)
    m8 = torch.ones((C, P, S), dtype=torch.uint8, device="cuda")
    timeit(f"apply_mask", lambda: ops.apply_mask(sv, m8), n, 2 * b + 1)
    timeit(f"mask_and", lambda: ops.mask_and(m8, m8), n, 3)
    del m8
    # value-window pooling and the median variants on subsets (O(window) per sample)
    Ps = min(P, 2000)
    svs, rgs = sv[:1, :Ps].contiguous(), rng[:1, :Ps].contiguous()
    ns = svs.numel()
    nvalid, bad = ops.range_rows_check(rgs)
    timeit(f"range_rows_check (1 x {Ps} x {S})", lambda: ops.range_rows_check(rgs), ns, b)
    timeit(f"pool_sv_value nanmean n=25 +-10 m (1 x {Ps} x {S})", lambda: ops.pool_sv_value(svs, rgs, nvalid, 10.0, 25, 20.0, lo, hi, threshold=12.0, want_pooled=False), ns, 0)
    timeit(f"pool_sv_value nanmean, every window summed (1 x {Ps} x {S})", lambda: ops.pool_sv_value(svs, rgs, nvalid, 10.0, 25, 20.0, lo, hi, threshold=12.0, want_pooled=False, running_sums=False), ns, 0)
    nv_all, _ = ops.range_rows_check(rng)
    timeit(f"pool_sv_value nanmean n=25 +-10 m ({C} x {P} x {S})", lambda: ops.pool_sv_value(sv, rng, nv_all, 10.0, 25, 20.0, lo, hi, threshold=12.0, want_pooled=False), n, 2 * b + 1)
    rng1 = rng[:, :1].expand(C, P, S).contiguous()  # one range vector for all pings of a channel (constant sound speed)
    nv1, _ = ops.range_rows_check(rng1)
    timeit(f"pool_sv_value nanmean, one range vector per channel ({C} x {P} x {S})", lambda: ops.pool_sv_value(sv, rng1, nv1, 10.0, 25, 20.0, lo, hi, threshold=12.0, want_pooled=False), n, 2 * b + 1)
    del rng1
    Pm = min(P, 64)
    svm, rgm = sv[:1, :Pm].contiguous(), rng[:1, :Pm].contiguous()
    nvm, _ = ops.range_rows_check(rgm)
    timeit(f"pool_sv nanmedian 51 x {2*n10+1} (1 x {Pm} x {S})", lambda: ops.pool_sv(svm, 100, 25, n10, func="nanmedian", threshold=12.0, want_pooled=False), svm.numel(), 0, reps=1)
    Pb = min(P, 4096)
    svb = sv[:1, :Pb].contiguous()
    timeit(f"pool_sv nanmedian 51 x {2*n10+1} (1 x {Pb} x {S})", lambda: ops.pool_sv(svb, 100, 25, n10, func="nanmedian", threshold=12.0, want_pooled=False), svb.numel(), 0, reps=1)
    del svb
    timeit(f"pool_sv_value nanmedian n=25 +-10 m (1 x {Pm} x {S})", lambda: ops.pool_sv_value(svm, rgm, nvm, 10.0, 25, 20.0, lo, hi, func="nanmedian", threshold=12.0, want_pooled=False), svm.numel(), 0, reps=1)
    Pb = min(P, 4096)
    svb = sv[:1, :Pb].contiguous()
    rgb = rng[:1, :1].expand(1, Pb, S).contiguous()
    nvb, _ = ops.range_rows_check(rgb)
    timeit(f"pool_sv_value nanmedian n=25 +-10 m, one range vector (1 x {Pb} x {S})", lambda: ops.pool_sv_value(svb, rgb, nvb, 10.0, 25, 20.0, lo, hi, func="nanmedian", threshold=12.0, want_pooled=False), svb.numel(), 0, reps=1)
    Pc = min(P, 256)
    timeit(f"pool_sv_value nanmedian, every window from memory (1 x {Pc} x {S})", lambda: ops.pool_sv_value(svb[:, :Pc].contiguous(), rgb[:, :Pc].contiguous(), nvb[:, :Pc].contiguous(), 10.0, 25, 20.0, lo, hi, func="nanmedian", threshold=12.0, want_pooled=False, running_sums=False), Pc * S, 0, reps=1)
    del sv, rng, svb, rgb
