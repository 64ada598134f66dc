"""Value-window pooling on a volume whose range rows differ from ping to ping -- for rocprofv3 --kernel-trace --stats."""
import sys
import torch
sys.path.insert(0, ".")
from echopype_amd import ops, synth

C, P, S = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (4, 20000, 2000)))
dt = torch.float32 if "f32" in sys.argv else torch.float64
d = synth.ek60_device(C, P, S)
cf = ops.power_coef_ek(d["sample_interval"], d["transmit_duration_nominal"], d["transmit_power"],
    d["sound_speed_indicative"], d["absorption_indicative"], d["gain_correction"], d["sa_correction"],
    d["equivalent_beam_angle"], d["frequency_nominal"], d["transmit_duration_nominal"][:, 0].contiguous(),
    pulse_length=d["pulse_length"], gain_is_table=True, sa_is_table=True)
sv, rng = ops.sv_power(d["backscatter_r"], cf, dtype=dt)
if "heave" in sys.argv:  # depth = range + a per-ping offset of a few sample steps (a moving platform)
    rng = rng + (0.4 * torch.sin(torch.arange(P, device="cuda", dtype=dt) * 0.9))[None, :, None]
lo, hi = ops.nanminmax(rng)
nv, _ = ops.range_rows_check(rng)
t = ops.Timer()
for i in range(3):
    t.start()
    ops.pool_sv_value(sv, rng, nv, 10.0, 25, 20.0, lo, hi, threshold=12.0, want_pooled=False)
    t.stop()
    print(f"{dt} {C}x{P}x{S}: {t.elapsed_ms():.2f} ms", flush=True)
