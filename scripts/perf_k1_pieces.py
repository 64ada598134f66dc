"""K1 (compute_Sv / compute_TS on power samples): the one-piece-workgroup kernel against the strided-rows one, per dtype
and with / without the echo_range array and its statistics -- development aid (round 5).  Run twice:
    EPA_K1_PIECES=1 python scripts/perf_k1_pieces.py ;  EPA_K1_PIECES=0 python scripts/perf_k1_pieces.py"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, ".")
from echopype_amd import ops, synth
shapes = [(4, 500000, 2000), (4, 250000, 4096)]
t = ops.Timer()
for C, P, S in shapes:
    d = synth.ek60_device(C, P, S, ss_every=1)
    cf = ops.power_coef_ek(d["sample_interval"], d["transmit_duration_nominal"], d["transmit_power"],
        d["sound_speed_indicative"], d["absorption_indicative"], d["gain_correction"], d["sa_correction"],
        d["equivalent_beam_angle"], d["frequency_nominal"], d["transmit_duration_nominal"][:, 0].contiguous(),
        pulse_length=d["pulse_length"], gain_is_table=True, sa_is_table=True)
    n = C * P * S
    for dt, b in ((torch.float64, 8), (torch.float32, 4)):
        out = torch.empty((C, P, S), dtype=dt, device="cuda")
        rng = torch.empty((C, P, S), dtype=dt, device="cuda")
        for name, kw, bps in (("Sv", dict(want_range=False, out=out), 4 + b),
                              ("Sv + range stats", dict(want_range=False, out=out, want_range_stats=True), 4 + b),
                              ("Sv + echo_range", dict(want_range=True, out=out, range_out=rng), 4 + 2 * b),
                              ("TS", dict(want_range=False, out=out, cal_type="TS"), 4 + b)):
            fn = lambda: ops.sv_power(d["backscatter_r"], cf, dtype=dt, **kw)
            fn(); torch.cuda.synchronize(); ms = []
            for _ in range(7):
                t.start(); fn(); t.stop(); ms.append(t.elapsed_ms())
            m = float(np.median(ms))
            print(f"pieces={os.environ.get('EPA_K1_PIECES', '1')} {C}x{P}x{S} {str(dt):14s} {name:18s} {m:8.3f} ms {n/m/1e6:8.1f} Gsamp/s "
                  f"{n*bps/m/1e9:6.2f} TB/s = {n*bps/m/1e9/8:.3f} of 8 TB/s", flush=True)
        del out, rng
    del d, cf
    torch.cuda.empty_cache()
