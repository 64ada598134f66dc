"""EPA_FUSED_STAGGER sweep on one stream of cfg5-shaped tiles and on the cfg2 volume (development probe)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, ".")
from echopype_amd import ops, synth, sharding
t = ops.Timer()
for (C, P, S, N) in ((4, 250000, 4096, 2), (4, 500000, 2000, 1)):
    sets = []
    dt = torch.float64
    for i in range(N):
        d = synth.ek60_device(C, P, S, seed=20260509 + i, ss_every=1)
        coef = ops.power_coef_ek(d["sample_interval"], d["transmit_duration_nominal"], d["transmit_power"],
            d["sound_speed_indicative"], d["absorption_indicative"], d["gain_correction"], d["sa_correction"],
            d["equivalent_beam_angle"], d["frequency_nominal"], d["transmit_duration_nominal"][:, 0].contiguous(),
            pulse_length=d["pulse_length"], gain_is_table=True, sa_is_table=True)
        ns = d["ping_time_ns"]; bin_ns = 20_000_000_000
        e0, _ = sharding.global_time_grid(ns.cpu().numpy(), bin_ns)
        n_t = P // 20
        bs = ops.time_bin_offsets(ns, e0, bin_ns, n_t)
        r_max = float((S - 1) * 2.56e-4 * float(d["sound_speed_indicative"].max()) / 2)
        n_r = len(np.arange(0, r_max + 1.0, 1.0)) - 1
        sv = torch.empty((C, P, S), dtype=dt, device="cuda"); mv = torch.empty((C, n_t, n_r), dtype=dt, device="cuda")
        sets.append((d["backscatter_r"], coef, bs, n_t, n_r, sv, mv)); del d
    for naps in (0, 8, 16, 32, 64, 128, 0, 32):
        os.environ["EPA_FUSED_STAGGER"] = str(naps)
        def go():
            for _ in range(4):
                for raw, coef, bs, n_t, n_r, sv, mv in sets:
                    ops.sv_mvbs_fused(raw, coef, bs, n_t, 1.0, n_r, dtype=dt, sv_out=sv, mvbs_out=mv)
        go(); torch.cuda.synchronize()
        ms = []
        for _ in range(3):
            t.start(); go(); t.stop(); ms.append(t.elapsed_ms() / (4 * N))
        m = float(np.median(ms))
        print(f"{C}x{P}x{S}  EPA_FUSED_STAGGER={naps:3d}  {m:7.3f} ms per launch  {C*P*S*12/m/1e9/8:.3f} of 8 TB/s", flush=True)
    del sets; torch.cuda.empty_cache()
