"""Chain pass 1 (Sv + noise estimate) against the fused Sv -> MVBS kernel on the same volume: same traffic
(4 B in, 8 B out per sample) -- development aid."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from echopype_amd import _lib, ops, synth
C, P, S = 4, int(sys.argv[1]) if len(sys.argv) > 1 else 250000, 2000
d = synth.ek60_device(C, P, S)
coef = ops.power_coef_ek(d["sample_interval"], d["transmit_duration_nominal"], d["transmit_power"],
    d["sound_speed_indicative"], d["absorption_indicative"], d["gain_correction"], d["sa_correction"],
    d["equivalent_beam_angle"], d["frequency_nominal"], d["transmit_duration_nominal"][:, 0].contiguous(),
    pulse_length=d["pulse_length"], gain_is_table=True, sa_is_table=True)
raw = d["backscatter_r"]
a2 = coef[..., _lib.CF_ALPHA2].contiguous()
ns = d["ping_time_ns"]; t0 = int(ns[0].item()); dtb = 20_000_000_000
n_t = P // 20
bs = ops.time_bin_offsets(ns, t0, dtb, n_t)
n_r = len(np.arange(0, float((S - 1) * 2.56e-4 * 1500.5 / 2) + 1.0, 1.0)) - 1
n = C * P * S
t = ops.Timer()
def timeit(name, fn, reps=5):
    fn(); torch.cuda.synchronize(); ms = []
    for _ in range(reps):
        t.start(); fn(); t.stop(); ms.append(t.elapsed_ms())
    m = float(np.median(ms))
    print(f"{name:44s} {m:8.3f} ms  {n*12/m/1e9:5.2f} TB/s", flush=True)
for rsn in (50, 2000):
    timeit(f"pass 1, noise blocks 20 x {rsn}", lambda: ops.sv_noise_fused(raw, coef, a2, 20, rsn))
timeit("pass 1, noise blocks 100 x 50", lambda: ops.sv_noise_fused(raw, coef, a2, 100, 50))
timeit("pass 1, 20 x 50, no Sv store", lambda: ops.sv_noise_fused(raw, coef, a2, 20, 50, want_sv=False))
timeit("pass 1, 20 x 50, + range statistics", lambda: ops.sv_noise_fused(raw, coef, a2, 20, 50, want_range_stats=True))
timeit("pass 1, 20 x 50, float32", lambda: ops.sv_noise_fused(raw, coef, a2, 20, 50, dtype=torch.float32))
timeit("pass 1, 20 x 50, float32, no Sv store", lambda: ops.sv_noise_fused(raw, coef, a2, 20, 50, dtype=torch.float32, want_sv=False))
timeit("fused Sv -> MVBS (20 s x 1 m), bins only", lambda: ops.sv_mvbs_fused(raw, coef, bs, n_t, 1.0, n_r, want_sv=False))
timeit("K1 compute_Sv alone", lambda: ops.sv_power(raw, coef, want_range=False))
timeit("fused Sv -> MVBS (20 s x 1 m)", lambda: ops.sv_mvbs_fused(raw, coef, bs, n_t, 1.0, n_r))
for nb in (20, 40, 100, 500):
    dtb2 = nb * 1_000_000_000
    n_t2 = P // nb
    bs2 = ops.time_bin_offsets(ns, t0, dtb2, n_t2)
    timeit(f"fused Sv -> MVBS ({nb} s x 1 m)", lambda: ops.sv_mvbs_fused(raw, coef, bs2, n_t2, 1.0, n_r))
