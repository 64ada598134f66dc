"""Does the fused Sv -> MVBS kernel stream faster as several concurrent launches?  (development probe)
The 2-rank gloo dry run on one GPU -- two processes, one kernel each, side by side -- moved 2 tiles in 15.9 ms where one
process needs 2 x 9.56 ms.  Here: N datasets of 4 x (500000 / N) x 2000, launched back to back on one stream or each on
its own stream."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from echopype_amd import ops, synth, sharding

C, PT, S = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (4, 500000, 2000)))
CARVE = len(sys.argv) > 4 and sys.argv[4] == "carve"   # the N datasets as consecutive blocks of ONE allocation
t = ops.Timer()
for dt in (torch.float64,):
    for N in (1, 2, 4):
        P = PT // N
        sets = []
        big_raw = torch.empty((N, C, P, S), dtype=torch.float32, device="cuda") if CARVE else None
        big_sv = torch.empty((N, C, P, S), dtype=dt, device="cuda") if CARVE else None
        for i in range(N):
            d = synth.ek60_device(C, P, S, seed=20260509 + i, ss_every=1)
            coef = ops.power_coef_ek(d["sample_interval"], d["transmit_duration_nominal"], d["transmit_power"],
                d["sound_speed_indicative"], d["absorption_indicative"], d["gain_correction"], d["sa_correction"],
                d["equivalent_beam_angle"], d["frequency_nominal"], d["transmit_duration_nominal"][:, 0].contiguous(),
                pulse_length=d["pulse_length"], gain_is_table=True, sa_is_table=True)
            ns = d["ping_time_ns"]
            bin_ns = 20_000_000_000
            e0, _ = sharding.global_time_grid(ns.cpu().numpy(), bin_ns)
            n_t = P // 20
            bs = ops.time_bin_offsets(ns, e0, bin_ns, n_t)
            r_max = float((S - 1) * 2.56e-4 * float(d["sound_speed_indicative"].max()) / 2)
            n_r = len(np.arange(0, r_max + 1.0, 1.0)) - 1
            sv = big_sv[i] if CARVE else torch.empty((C, P, S), dtype=dt, device="cuda")
            mv = torch.empty((C, n_t, n_r), dtype=dt, device="cuda")
            raw = d["backscatter_r"]
            if CARVE:
                big_raw[i].copy_(raw)
                raw = big_raw[i]
            sets.append((raw, coef, bs, n_t, n_r, sv, mv))
            del d
        streams = [torch.cuda.Stream() for _ in range(N)]
        def seq():
            for raw, coef, bs, n_t, n_r, sv, mv in sets:
                ops.sv_mvbs_fused(raw, coef, bs, n_t, 1.0, n_r, dtype=dt, sv_out=sv, mvbs_out=mv)
        def conc():
            cur = torch.cuda.current_stream()
            for st, (raw, coef, bs, n_t, n_r, sv, mv) in zip(streams, sets):
                st.wait_stream(cur)
                with torch.cuda.stream(st):
                    ops.sv_mvbs_fused(raw, coef, bs, n_t, 1.0, n_r, dtype=dt, sv_out=sv, mvbs_out=mv)
            for st in streams:
                cur.wait_stream(st)
        for name, fn in (("one stream", seq), ("own streams", conc)):
            if N == 1 and name == "own streams":
                continue
            fn(); torch.cuda.synchronize()
            ms = []
            for _ in range(5):
                t.start(); fn(); t.stop(); ms.append(t.elapsed_ms())
            m = float(np.median(ms))
            b = 12 if dt == torch.float64 else 8
            print(f"{str(dt):14s} {N} x 4x{P}x{S}  {name:11s} {m:8.3f} ms  {C*PT*S/m/1e6:7.1f} Gsamp/s  {C*PT*S*b/m/1e9:5.2f} TB/s = {C*PT*S*b/m/1e9/8:.3f}", flush=True)
        del sets, big_raw, big_sv
        torch.cuda.empty_cache()
