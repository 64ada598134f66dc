"""Quick kernel timing probe (HIP events) -- development aid, not the bench."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
from echopype_amd import ops, synth

C, P, S = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (4, 100000, 2000)))
d = synth.ek60_device(C, P, S)
torch.cuda.synchronize()
f64 = torch.float64
def coef():
    return ops.power_coef_ek(d["sample_interval"], d["transmit_duration_nominal"], d["transmit_power"],
        d["sound_speed_indicative"], d["absorption_indicative"], d["gain_correction"], d["sa_correction"],
        d["equivalent_beam_angle"], d["frequency_nominal"], d["transmit_duration_nominal"][:, 0].contiguous(),
        pulse_length=d["pulse_length"], gain_is_table=True, sa_is_table=True)
cf = coef()
n = C * P * S
t = ops.Timer()
def timeit(name, fn, bytes_per_sample, reps=5):
    fn(); torch.cuda.synchronize()
    ms = []
    for _ in range(reps):
        t.start(); fn(); t.stop(); ms.append(t.elapsed_ms())
    m = float(np.median(ms))
    print(f"{name:40s} {m:8.3f} ms  {n/m/1e6:8.1f} Gsamp/s  {n*bytes_per_sample/m/1e9:7.2f} TB/s (algorithmic)", flush=True)
for dt, b in ((torch.float64, 8), (torch.float32, 4)):
    out = torch.empty((C, P, S), dtype=dt, device="cuda")
    rng = torch.empty((C, P, S), dtype=dt, device="cuda")
    timeit(f"sv_power {dt} (no range)", lambda: ops.sv_power(d["backscatter_r"], cf, dtype=dt, want_range=False, out=out), 4 + b)
    timeit(f"sv_power {dt} (+range)", lambda: ops.sv_power(d["backscatter_r"], cf, dtype=dt, out=out, range_out=rng), 4 + 2 * b)
    ns = d["ping_time_ns"]
    t0 = int(ns[0].item()); dtb = 20_000_000_000
    n_t = int((int(ns[-1].item()) - t0) // dtb) + 1
    bs = ops.time_bin_offsets(ns, t0, dtb, n_t)
    n_r = int(np.ceil(S * 2.56e-4 * 1500.5 / 2)) + 1
    mv = torch.empty((C, n_t, n_r), dtype=dt, device="cuda")
    timeit(f"fused Sv+MVBS {dt} (write Sv)", lambda: ops.sv_mvbs_fused(d["backscatter_r"], cf, bs, n_t, 1.0, n_r, dtype=dt, sv_out=out, mvbs_out=mv), 4 + b)
    timeit(f"fused Sv+MVBS {dt} (MVBS only)", lambda: ops.sv_mvbs_fused(d["backscatter_r"], cf, bs, n_t, 1.0, n_r, dtype=dt, want_sv=False, mvbs_out=mv), 4)
    timeit(f"mvbs standalone {dt} (range array)", lambda: ops.mvbs(out, bs, n_t, 1.0, n_r, range=rng), 2 * b)
    timeit(f"mvbs standalone {dt} (affine)", lambda: ops.mvbs(out, bs, n_t, 1.0, n_r, coef=cf), b)
    a2 = (2 * d["absorption_indicative"]).contiguous()
    nb = ops.noise_estimate(out, a2, 20, 50, range=rng)
    timeit(f"noise_estimate {dt}", lambda: ops.noise_estimate(out, a2, 20, 50, range=rng), 2 * b)
    sn = torch.empty_like(out); 
    timeit(f"noise_apply {dt}", lambda: ops.noise_apply(out, a2, nb, 20, 3.0, range=rng), 4 * b)
    del out, rng, sn
timeit("power_coef_ek", coef, 0)
