"""Timing probe of the fused Sv->MVBS kernel only (development aid)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, ".")
from echopype_amd import ops, synth
C, P, S = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (4, 100000, 2000)))
d = synth.ek60_device(C, P, S)
cf = ops.power_coef_ek(d["sample_interval"], d["transmit_duration_nominal"], d["transmit_power"],
    d["sound_speed_indicative"], d["absorption_indicative"], d["gain_correction"], d["sa_correction"],
    d["equivalent_beam_angle"], d["frequency_nominal"], d["transmit_duration_nominal"][:, 0].contiguous(),
    pulse_length=d["pulse_length"], gain_is_table=True, sa_is_table=True)
n = C * P * S
t = ops.Timer()
ns = d["ping_time_ns"]; t0 = int(ns[0].item()); dtb = 20_000_000_000
n_t = int((int(ns[-1].item()) - t0) // dtb) + 1
bs = ops.time_bin_offsets(ns, t0, dtb, n_t)
n_r = int(np.ceil(S * 2.56e-4 * 1500.5 / 2)) + 1
for dt, b in ((torch.float64, 8), (torch.float32, 4)):
    out = torch.empty((C, P, S), dtype=dt, device="cuda")
    mv = torch.empty((C, n_t, n_r), dtype=dt, device="cuda")
    for name, kw, bps in (("write Sv", dict(sv_out=out), 4 + b), ("Sv + stats", dict(sv_out=out, want_range_stats=True), 4 + b),
                          ("MVBS only", dict(want_sv=False), 4)):
        fn = lambda: ops.sv_mvbs_fused(d["backscatter_r"], cf, bs, n_t, 1.0, n_r, dtype=dt, mvbs_out=mv, **kw)
        fn(); torch.cuda.synchronize(); ms = []
        for _ in range(7):
            t.start(); fn(); t.stop(); ms.append(t.elapsed_ms())
        m = float(np.median(ms))
        print(f"VEC={os.environ.get('EPA_REDUCE_VEC','dflt')} fused {str(dt):14s} {name:10s} {m:8.3f} ms {n/m/1e6:8.1f} Gsamp/s {n*bps/m/1e9:6.2f} TB/s", flush=True)
    del out, mv
