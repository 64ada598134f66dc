"""Development aid: pool_sv_value nanmean on 4 x 100 000 x 2000 with one range vector per channel (run under
rocprofv3 --kernel-trace --stats for the per-kernel split)."""
import sys
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
del rng
nv1, _ = ops.range_rows_check(rng1)
t = ops.Timer()
for _ in range(3):
    t.start(); ops.pool_sv_value(sv, rng1, nv1, 10.0, 25, 20.0, lo, hi, threshold=12.0, want_pooled=False); t.stop()
    print("ms", t.elapsed_ms())
