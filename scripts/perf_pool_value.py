"""Timing probe of the value-window nanmean pooling (mask_transient_noise on ``depth``) -- development aid.
    python scripts/perf_pool_value.py [C P S]
Rows: the range vector of a channel changes every 2000 pings / at every ping / never."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from echopype_amd import ops, synth

C, P, S = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (4, 100000, 2000)))
t = ops.Timer()
import os
for ss in ([int(x) for x in os.environ['PV_SS'].split(',')] if os.environ.get('PV_SS') else (2000, 1, P)):
    d = synth.ek60_device(C, P, S, ss_every=ss)
    cf = ops.power_coef_ek(d["sample_interval"], d["transmit_duration_nominal"], d["transmit_power"],
        d["sound_speed_indicative"], d["absorption_indicative"], d["gain_correction"], d["sa_correction"],
        d["equivalent_beam_angle"], d["frequency_nominal"], d["transmit_duration_nominal"][:, 0].contiguous(),
        pulse_length=d["pulse_length"], gain_is_table=True, sa_is_table=True)
    for dt in (torch.float64, torch.float32):
        sv, rng = ops.sv_power(d["backscatter_r"], cf, dtype=dt)
        lo, hi = ops.nanminmax(rng)
        nv, _ = ops.range_rows_check(rng)
        fn = lambda: ops.pool_sv_value(sv, rng, nv, 10.0, 25, 20.0, lo, hi, threshold=12.0, want_pooled=False)
        m0 = fn()[1]
        torch.cuda.synchronize()
        ms = []
        for _ in range(3):
            t.start(); fn(); t.stop(); ms.append(t.elapsed_ms())
        m = float(np.median(ms))
        print(f"range vector changes every {ss:6d} pings  {str(dt):14s} {C}x{P}x{S}  {m:9.3f} ms  {sv.numel()/m/1e6:8.2f} Gsamp/s"
              f"  mask sum {int(m0.sum())}", flush=True)
        del sv, rng, m0
    del d
