"""Timing probe of epa_nasc (HIP events) -- development aid."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from echopype_amd import ops, synth
C, P, S = 4, 100000, 2000
d = synth.ek60_device(C, P, S)
cf = ops.power_coef_ek(d["sample_interval"], d["transmit_duration_nominal"], d["transmit_power"],
    d["sound_speed_indicative"], d["absorption_indicative"], d["gain_correction"], d["sa_correction"],
    d["equivalent_beam_angle"], d["frequency_nominal"], d["transmit_duration_nominal"][:, 0].contiguous(),
    pulse_length=d["pulse_length"], gain_is_table=True, sa_is_table=True)
t = ops.Timer()
for dt, b in ((torch.float64, 8), (torch.float32, 4)):
    sv, rng = ops.sv_power(d["backscatter_r"], cf, dtype=dt)
    lo, hi = ops.nanminmax(rng)
    n_r = len(np.arange(0, hi + 10.0, 10.0)) - 1
    for n_d in (556, 20):   # 100 000 pings at 1 Hz and 10 kn = 278 nmi -> 556 bins of 0.5 nmi; and a short leg
        starts = torch.from_numpy(np.linspace(0, P, n_d + 1).astype(np.int32)).cuda()
        ms = []
        for _ in range(4):
            t.start(); ops.nasc(sv, rng, starts, n_d, 10.0, n_r); t.stop(); ms.append(t.elapsed_ms())
        m = float(np.median(ms[1:])); n = sv.numel()
        print(f"nasc {dt} {n_d:4d} distance bins x {n_r} depth bins: {m:7.3f} ms  {n/m/1e6:7.1f} Gsamp/s  {n*2*b/m/1e9:5.2f} TB/s (algorithmic)", flush=True)
