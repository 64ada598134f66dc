"""Only the value-window transient pooling on a channel whose pings share one range vector (the reference's DEFAULT
mask_transient_noise on the usual echo_range), for a rocprofv3 kernel trace -- development aid."""
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
sv, rng = ops.sv_power(d["backscatter_r"], cf, dtype=torch.float64)
lo, hi = ops.nanminmax(rng)
rng1 = rng[:, :1].expand(C, P, S).contiguous()
nv1, _ = ops.range_rows_check(rng1)
t = ops.Timer()
for i in range(4):
    t.start()
    ops.pool_sv_value(sv, rng1, nv1, 10.0, 25, 20.0, lo, hi, threshold=12.0, want_pooled=False)
    t.stop()
    print("call", i, round(t.elapsed_ms(), 3), "ms", flush=True)
