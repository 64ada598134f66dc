"""compute_MVBS on an existing Sv through the coefficient rows (epa_mvbs with coef): timing probe at the cfg2 volume
(development aid).  python scripts/perf_mvbs_rows.py [ss_every]"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from echopype_amd import ops, synth

ss = int(sys.argv[1]) if len(sys.argv) > 1 else 1
C, P, S = 4, 500_000, 2000
d = synth.ek60_device(C, P, S, ss_every=ss)
tau0 = d["transmit_duration_nominal"][:, 0].contiguous()
coef = ops.power_coef_ek(d["sample_interval"], d["transmit_duration_nominal"], d["transmit_power"], d["sound_speed_indicative"],
                         d["absorption_indicative"], d["gain_correction"], d["sa_correction"], d["equivalent_beam_angle"],
                         d["frequency_nominal"], tau0, pulse_length=d["pulse_length"], gain_is_table=True, sa_is_table=True)
ns = d["ping_time_ns"]
n_t = P // 20
bs = ops.time_bin_offsets(ns, int(ns[0].item()), 20_000_000_000, n_t)
n_r = len(np.arange(0, float((S - 1) * 2.56e-4 * 1500.5 / 2) + 1.0, 1.0)) - 1
t = ops.Timer()
for dt in (torch.float64, torch.float32):
    sv, _ = ops.sv_power(d["backscatter_r"], coef, dtype=dt, want_range=False)
    ms = []
    for _ in range(9):
        t.start(); r = ops.mvbs(sv, bs, n_t, 1.0, n_r, coef=coef, coef_as_stored=True); t.stop(); ms.append(t.elapsed_ms())
    m = float(np.median(ms[2:]))
    esz = 8 if dt == torch.float64 else 4
    print(f"ss_every={ss} {str(dt)[6:]}: {m:.3f} ms  {C*P*S*esz/m/1e9:.2f} TB/s", flush=True)
    del sv, r
