"""compute_Sv -> remove_background_noise -> compute_MVBS(Sv_corrected): four separate kernels vs the
two-pass chain (HIP events) -- development aid."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from echopype_amd import _lib, ops, synth
C, P, S = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (4, 250000, 2000)))
d = synth.ek60_device(C, P, S)
coef = ops.power_coef_ek(d["sample_interval"], d["transmit_duration_nominal"], d["transmit_power"],
    d["sound_speed_indicative"], d["absorption_indicative"], d["gain_correction"], d["sa_correction"],
    d["equivalent_beam_angle"], d["frequency_nominal"], d["transmit_duration_nominal"][:, 0].contiguous(),
    pulse_length=d["pulse_length"], gain_is_table=True, sa_is_table=True)
raw = d["backscatter_r"]
a2 = coef[..., _lib.CF_ALPHA2].contiguous()
ns = d["ping_time_ns"]; t0 = int(ns[0].item()); dtb = 20_000_000_000
n_t = int((int(ns[-1].item()) - t0) // dtb) + 1
bs = ops.time_bin_offsets(ns, t0, dtb, n_t)
n = C * P * S
t = ops.Timer()
def timeit(name, fn, bytes_per_sample, reps=4):
    fn(); torch.cuda.synchronize(); ms = []
    for _ in range(reps):
        t.start(); fn(); t.stop(); ms.append(t.elapsed_ms())
    m = float(np.median(ms))
    print(f"{name:58s} {m:8.3f} ms  {n/m/1e6:7.1f} Gsamp/s  {n*bytes_per_sample/m/1e9:5.2f} TB/s (algorithmic)", flush=True)
    return m
for dt, b in ((torch.float64, 8), (torch.float32, 4)):
    print(f"-- {dt}, {C} x {P} x {S}", flush=True)
    sv, rng = ops.sv_power(raw, coef, dtype=dt)
    _, rmax = ops.nanminmax(rng)
    n_r = len(np.arange(0, rmax + 1.0, 1.0)) - 1
    noise = ops.noise_estimate(sv, a2, 20, 50, range=rng)
    sn = torch.empty_like(sv); sc = torch.empty_like(sv)
    def four():
        ops.sv_power(raw, coef, dtype=dt, out=sv, range_out=rng)
        nz = ops.noise_estimate(sv, a2, 20, 50, range=rng)
        a, c = ops.noise_apply(sv, a2, nz, 20, 3.0, range=rng)
        return ops.mvbs(c, bs, n_t, 1.0, n_r, range=rng)
    m4 = timeit("K1(+range) + K6 + K7 + K5  [4 + 5b + ...]", four, 4 + 2 * b + 2 * b + 4 * b + 2 * b)
    del sn, sc
    def two(noise_out=True):
        s1, _, nz, rm = ops.sv_noise_fused(raw, coef, a2, 20, 50, dtype=dt, want_range_max=True)
        return ops.sv_denoise_mvbs(raw, coef, a2, nz, 20, 3.0, bs, n_t, 1.0, n_r, dtype=dt, want_noise=noise_out)
    m2 = timeit("two passes: Sv, Sv_noise, Sv_corrected, MVBS", lambda: two(True), 8 + 3 * b)
    m2b = timeit("two passes: Sv, Sv_corrected, MVBS", lambda: two(False), 8 + 2 * b)
    timeit("  pass 1 alone (raw -> Sv + noise estimate)", lambda: ops.sv_noise_fused(raw, coef, a2, 20, 50, dtype=dt), 4 + b)
    timeit("  pass 2 alone (raw -> Sv_corrected + MVBS)", lambda: ops.sv_denoise_mvbs(raw, coef, a2, noise, 20, 3.0, bs, n_t, 1.0, n_r, dtype=dt), 4 + b)
    print(f"   speed-up of the chain: {m4/m2:.2f}x with Sv_noise, {m4/m2b:.2f}x without", flush=True)
    del sv, rng
