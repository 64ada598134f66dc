"""Fused kernel: time of ONE launch against its size (development probe): 4 x P x S for P = 31250 .. 1 M."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from echopype_amd import ops, synth, sharding
t = ops.Timer()
dt = torch.float64
for S in (4096, 2000):
    PM = 1000000 if S == 4096 else 2000000
    d = synth.ek60_device(4, PM, S, seed=20260509, ss_every=1)
    coefM = ops.power_coef_ek(d["sample_interval"], d["transmit_duration_nominal"], d["transmit_power"],
        d["sound_speed_indicative"], d["absorption_indicative"], d["gain_correction"], d["sa_correction"],
        d["equivalent_beam_angle"], d["frequency_nominal"], d["transmit_duration_nominal"][:, 0].contiguous(),
        pulse_length=d["pulse_length"], gain_is_table=True, sa_is_table=True)
    svM = torch.empty((4, PM, S), dtype=dt, device="cuda")
    P = PM
    while P >= 31250:
        raw = d["backscatter_r"][:, :P].contiguous() if P < PM else d["backscatter_r"]
        coef = coefM[:, :P].contiguous()
        ns = d["ping_time_ns"][:P].contiguous()
        e0, _ = sharding.global_time_grid(ns.cpu().numpy(), 20_000_000_000)
        n_t = P // 20
        bs = ops.time_bin_offsets(ns, e0, 20_000_000_000, n_t)
        r_max = float((S - 1) * 2.56e-4 * float(d["sound_speed_indicative"].max()) / 2)
        n_r = len(np.arange(0, r_max + 1.0, 1.0)) - 1
        sv = svM.view(-1)[: 4 * P * S].view(4, P, S)
        mv = torch.empty((4, n_t, n_r), dtype=dt, device="cuda")
        fn = lambda: ops.sv_mvbs_fused(raw, coef, bs, n_t, 1.0, n_r, dtype=dt, sv_out=sv, mvbs_out=mv)
        fn(); torch.cuda.synchronize()
        ms = []
        for _ in range(5):
            t.start(); fn(); t.stop(); ms.append(t.elapsed_ms())
        m = float(np.median(ms))
        print(f"4x{P}x{S}  {m:8.3f} ms  {m / P * 250000:7.3f} ms per 250 000 pings  {4*P*S*12/m/1e9/8:.3f} of 8 TB/s", flush=True)
        del raw, coef, mv
        P //= 2
    del d, coefM, svM
    torch.cuda.empty_cache()
