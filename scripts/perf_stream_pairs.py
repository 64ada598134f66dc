"""Which PAIR of HIP streams runs two launches of the fused kernel side by side?  (round 6: the headline's two-stream gain
came and went between runs of one bench process.)  K streams are created and used once in order -- the runtime binds a
stream to one of GPU_MAX_HW_QUEUES hardware queues at its first use -- then two datasets of 4 x 250000 x 2000 are
launched on every pair (0, j) and a few others; a pair whose queues share a dispatch pipe serialises.
    GPU_MAX_HW_QUEUES=N python scripts/perf_stream_pairs.py [K]"""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from echopype_amd import ops, synth, sharding

K = int(sys.argv[1]) if len(sys.argv) > 1 else 10
C, P, S = 4, 250000, 2000
dt = torch.float64
sets = []
for i in range(2):
    d = synth.ek60_device(C, P, S, seed=20260509 + i, ss_every=1)
    coef = ops.power_coef_ek(d["sample_interval"], d["transmit_duration_nominal"], d["transmit_power"],
        d["sound_speed_indicative"], d["absorption_indicative"], d["gain_correction"], d["sa_correction"],
        d["equivalent_beam_angle"], d["frequency_nominal"], d["transmit_duration_nominal"][:, 0].contiguous(),
        pulse_length=d["pulse_length"], gain_is_table=True, sa_is_table=True)
    ns = d["ping_time_ns"]
    e0, _ = sharding.global_time_grid(ns.cpu().numpy(), 20_000_000_000)
    n_t = P // 20
    bs = ops.time_bin_offsets(ns, e0, 20_000_000_000, n_t)
    n_r = len(np.arange(0, float((S - 1) * 2.56e-4 * float(d["sound_speed_indicative"].max()) / 2) + 1.0, 1.0)) - 1
    sets.append((d["backscatter_r"], coef, bs, n_t, n_r, torch.empty((C, P, S), dtype=dt, device="cuda"),
                 torch.empty((C, n_t, n_r), dtype=dt, device="cuda")))
    del d
streams = [torch.cuda.Stream() for _ in range(K)]
for st in streams:  # first use, in order
    with torch.cuda.stream(st):
        torch.zeros(8, device="cuda").add_(1)
torch.cuda.synchronize()
t = ops.Timer()

def run(pair):
    cur = torch.cuda.current_stream()
    for st, (raw, coef, bs, n_t, n_r, sv, mv) in zip(pair, sets):
        st.wait_stream(cur)
        with torch.cuda.stream(st):
            ops.sv_mvbs_fused(raw, coef, bs, n_t, 1.0, n_r, dtype=dt, sv_out=sv, mvbs_out=mv)
    for st in pair:
        cur.wait_stream(st)

print("GPU_MAX_HW_QUEUES =", os.environ.get("GPU_MAX_HW_QUEUES"), " streams:", K, flush=True)
cur = torch.cuda.current_stream()
pairs = [(None, None)] + [(0, j) for j in range(1, K)] + [q for q in ((1, 5), (2, 6), (3, 7), (1, 2), (5, 6)) if max(q) < K]
for i, j in pairs:
    pair = (cur, cur) if i is None else (streams[i], streams[j])
    run(pair); torch.cuda.synchronize()
    ms = []
    for _ in range(5):
        t.start(); run(pair); t.stop(); ms.append(t.elapsed_ms())
    m = float(np.median(ms))
    n = 2 * C * P * S
    print(f"pair {('one stream' if i is None else (i, j))!s:12s} {m:7.3f} ms  {n * 12 / m / 1e9:5.2f} TB/s = {n * 12 / m / 1e9 / 8:.3f}", flush=True)
