"""compute_MVBS's kernel on an existing Sv (EK60 4 x 500 000 x 2000): range from the array vs from coefficient rows --
development aid (also the target of rocprofv3 --pmc runs)."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from echopype_amd import ops, synth, sharding

C, P, S = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (4, 500000, 2000)))
dt = torch.float32 if "f32" in sys.argv else torch.float64
d = synth.ek60_device(C, P, S)
cf = ops.power_coef_ek(d["sample_interval"], d["transmit_duration_nominal"], d["transmit_power"],
    d["sound_speed_indicative"], d["absorption_indicative"], d["gain_correction"], d["sa_correction"],
    d["equivalent_beam_angle"], d["frequency_nominal"], d["transmit_duration_nominal"][:, 0].contiguous(),
    pulse_length=d["pulse_length"], gain_is_table=True, sa_is_table=True)
sv, rng = ops.sv_power(d["backscatter_r"], cf, dtype=dt)
ns = d["ping_time_ns"]
e0, _ = sharding.global_time_grid(ns.cpu().numpy(), 20_000_000_000)
n_t = P // 20
bs = ops.time_bin_offsets(ns, e0, 20_000_000_000, n_t)
hi = ops.nanminmax(rng)[1]
n_r = len(np.arange(0, hi + 1.0, 1.0)) - 1
t = ops.Timer()
for name, kw in (("range array", dict(range=rng)), ("coefficient rows", dict(coef=cf, coef_as_stored=True))):
    ms = []
    for i in range(4):
        t.start(); r = ops.mvbs(sv, bs, n_t, 1.0, n_r, **kw); t.stop(); ms.append(t.elapsed_ms())
    m = float(np.median(ms[1:]))
    b = sv.element_size() * (2 if "range" in kw else 1)
    print(f"mvbs {dt} {name:18s} {m:7.2f} ms  {sv.numel()/m/1e6:7.1f} Gsamp/s  {sv.numel()*b/m/1e9:5.2f} TB/s", flush=True)
