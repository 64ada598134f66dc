"""The two chain passes as the API runs them (pass 1 with the echo_range statistics, pass 2 with Sv_noise, Sv_corrected
and their minima / maxima), HIP events, 4 x P x 2000 -- development aid for the statistics' atomics (round 6)."""
import sys
sys.path.insert(0, ".")
import numpy as np, torch
from echopype_amd import _lib, ops, synth
C, P, S = 4, int(sys.argv[1]) if len(sys.argv) > 1 else 250000, 2000
n = C * P * S
t = ops.Timer()
def timeit(name, fn, bps, reps=6):
    fn(); fn(); torch.cuda.synchronize(); ms = []
    for _ in range(reps):
        t.start(); fn(); t.stop(); ms.append(t.elapsed_ms())
    m = float(np.median(ms))
    print(f"{name:64s} {m:8.3f} ms  {n * bps / m / 1e9:5.2f} TB/s  frac {n * bps / m / 1e9 / 8:.3f}", flush=True)
for ss in (1, 2000):
    d = synth.ek60_device(C, P, S, ss_every=ss)
    coef = ops.power_coef_ek(d["sample_interval"], d["transmit_duration_nominal"], d["transmit_power"],
        d["sound_speed_indicative"], d["absorption_indicative"], d["gain_correction"], d["sa_correction"],
        d["equivalent_beam_angle"], d["frequency_nominal"], d["transmit_duration_nominal"][:, 0].contiguous(),
        pulse_length=d["pulse_length"], gain_is_table=True, sa_is_table=True)
    raw = d["backscatter_r"]
    a2 = coef[..., _lib.CF_ALPHA2].contiguous()
    ns = d["ping_time_ns"]; t0 = int(ns[0].item()); dtb = 20_000_000_000
    n_t = int((int(ns[-1].item()) - t0) // dtb) + 1
    bs = ops.time_bin_offsets(ns, t0, dtb, n_t)
    n_r = len(np.arange(0, float((S - 1) * 2.56e-4 * 1500.5 / 2) + 1.0, 1.0)) - 1
    for dt, b in ((torch.float64, 8), (torch.float32, 4)):
        out = ops.sv_noise_fused(raw, coef, a2, 20, 50, dtype=dt, want_range_stats=True)
        noise = out[2]
        tag = f"ss_every {ss} {str(dt)[6:]}"
        timeit(f"{tag} pass 1 with range statistics", lambda: ops.sv_noise_fused(raw, coef, a2, 20, 50, dtype=dt, want_range_stats=True), 4 + b)
        timeit(f"{tag} pass 1 without", lambda: ops.sv_noise_fused(raw, coef, a2, 20, 50, dtype=dt), 4 + b)
        with _lib.launch_trace() as tr:
            ops.sv_denoise_mvbs(raw, coef, a2, noise, 20, 3.0, bs, n_t, 1.0, n_r, dtype=dt, want_noise=True, want_minmax=True, minmax_async=True)
        print("   pass 2 kernels:", [k for k in tr.kernels if "denoise" in k])
        timeit(f"{tag} pass 2 with minima / maxima", lambda: ops.sv_denoise_mvbs(raw, coef, a2, noise, 20, 3.0, bs, n_t, 1.0, n_r, dtype=dt, want_noise=True, want_minmax=True, minmax_async=True), 4 + 2 * b)
        timeit(f"{tag} pass 2 without", lambda: ops.sv_denoise_mvbs(raw, coef, a2, noise, 20, 3.0, bs, n_t, 1.0, n_r, dtype=dt, want_noise=True), 4 + 2 * b)
    del d, raw, coef
