"""Two streams of fused-kernel launches over cfg5-shaped tiles: in lockstep or staggered by half a kernel (development probe)."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from echopype_amd import ops, synth, sharding

C, P, S, N = 4, 250000, 4096, 4
dt = torch.float64
sets = []
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
    sv = torch.empty((C, P, S), dtype=dt, device="cuda")
    mv = torch.empty((C, n_t, n_r), dtype=dt, device="cuda")
    sets.append((d["backscatter_r"], coef, bs, n_t, n_r, sv, mv))
    del d
pad = torch.empty(6_000_000_000, dtype=torch.float32, device="cuda")  # zeroing it: ~4.5 ms
t = ops.Timer()
REPS = 6
def run(k, stagger):
    streams = [torch.cuda.Stream() for _ in range(k)]
    cur = torch.cuda.current_stream()
    def go():
        for st in streams:
            st.wait_stream(cur)
        if stagger and k > 1:
            for j in range(1, k):
                with torch.cuda.stream(streams[j]):
                    pad[: pad.numel() * j // k].zero_()
        for r in range(REPS):
            for i, (raw, coef, bs, n_t, n_r, sv, mv) in enumerate(sets):
                with torch.cuda.stream(streams[(r * N + i) % k]):
                    ops.sv_mvbs_fused(raw, coef, bs, n_t, 1.0, n_r, dtype=dt, sv_out=sv, mvbs_out=mv)
        for st in streams:
            cur.wait_stream(st)
    go(); torch.cuda.synchronize()
    ms = []
    for _ in range(3):
        t.start(); go(); t.stop(); ms.append(t.elapsed_ms())
    m = float(np.median(ms)) / (REPS * N)
    print(f"{k} stream(s) {'staggered' if stagger else 'lockstep ':9s}  {m:7.3f} ms per tile  {C*P*S*12/m/1e9/8:.3f} of 8 TB/s   (runs: {[round(x/(REPS*N),3) for x in ms]})", flush=True)
for k, stg in ((1, False), (2, False), (2, True), (3, False), (3, True), (2, False), (2, True), (1, False)):
    run(k, stg)
