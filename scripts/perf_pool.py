"""A/B probe of the pooled-Sv box kernels (development aid)."""
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
sv, _ = ops.sv_power(d["backscatter_r"], cf, dtype=torch.float64, want_range=False)
t = ops.Timer()
for n, m in ((25, 53), (2, 5)):
    ms = []
    for _ in range(4):
        t.start(); ops.pool_sv(sv, 100, n, m, threshold=12.0, want_pooled=False); t.stop(); ms.append(t.elapsed_ms())
    print(f"window {2*n+1} x {2*m+1}: {np.median(ms[1:]):.2f} ms", flush=True)
