"""Development aid: attenuated_mask on 4 x 100 000 x 2000 (run under rocprofv3 --kernel-trace --stats for the split)."""
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
for dt in (torch.float64, torch.float32):
    sv, rng = ops.sv_power(d["backscatter_r"], cf, dtype=dt)
    t = ops.Timer()
    for _ in range(3):
        t.start(); m = ops.attenuated_mask(sv, rng, 150.0, 250.0, 15, -6.0); t.stop()
    print(dt, "ms", t.elapsed_ms(), "flagged pings", int(m[:, :, 0].sum()))
