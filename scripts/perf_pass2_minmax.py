"""Pass 2 of the chain (raw -> Sv_noise, Sv_corrected, MVBS) with and without the actual_range by-product, a new sound
speed at every ping (the drift kernel) -- development aid."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from echopype_amd import ops, synth
C, P, S = 4, 500_000, 2000
d = synth.ek60_device(C, P, S, ss_every=1)
coef = ops.power_coef_ek(d["sample_interval"], d["transmit_duration_nominal"], d["transmit_power"],
                         d["sound_speed_indicative"], d["absorption_indicative"], d["gain_correction"], d["sa_correction"],
                         d["equivalent_beam_angle"], d["frequency_nominal"], d["transmit_duration_nominal"][:, 0].contiguous(),
                         pulse_length=d["pulse_length"], gain_is_table=True, sa_is_table=True)
a2 = coef[..., 4].contiguous()
n_t = P // 20
bs = ops.time_bin_offsets(d["ping_time_ns"], int(d["ping_time_ns"][0].item()), 20_000_000_000, n_t)
n_r = len(np.arange(0, float((S - 1) * 2.56e-4 * 1500.5 / 2) + 1.0, 1.0)) - 1
sv, _, nz = ops.sv_noise_fused(d["backscatter_r"], coef, a2, 20, 50)
del sv
t = ops.Timer()
bs1 = torch.arange(0, P + 20, 20, dtype=torch.int32, device="cuda").clamp_(max=P)
for name, kw in (("MVBS 1 m bins", dict(bin_start=bs, n_t=n_t, rb=1.0, n_r=n_r)), ("one range bin", dict(bin_start=bs1, n_t=bs1.numel() - 1, rb=1e30, n_r=1))):
    for mm in (False, True):
        fn = lambda: ops.sv_denoise_mvbs(d["backscatter_r"], coef, a2, nz, 20, 3.0, kw["bin_start"], kw["n_t"], kw["rb"], kw["n_r"],  # noqa: E731
                                         want_noise=True, want_minmax=mm)
        r = fn(); torch.cuda.synchronize(); del r; ms = []
        for _ in range(5):
            t.start(); r = fn(); t.stop(); ms.append(t.elapsed_ms()); del r
        print(f"{name:14s} minmax={mm!s:5s} {np.median(ms):7.2f} ms", flush=True)
