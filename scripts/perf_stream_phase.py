"""Does the two-stream gain depend on the RELATIVE position of the two launches' walks?  (round 6: it moves 0.66-0.75 from
process to process with every stream pair alike.)  Two one-channel datasets of 1 x 1 000 000 x 2000; kernel B is given a
view of its dataset that starts ``d`` pings in (same length for every d), so the two walks -- in lockstep otherwise --
are offset by d x 8 KB (raw) / d x 16 KB (Sv)."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from echopype_amd import ops, synth, sharding

C, P, S = 1, 1_000_000, 2000
L = 800_000
dt = torch.float64
sets = []
for i in range(2):
    d = synth.ek60_device(C, P, S, seed=20260509 + i, ss_every=1)
    coef = ops.power_coef_ek(d["sample_interval"], d["transmit_duration_nominal"], d["transmit_power"],
        d["sound_speed_indicative"], d["absorption_indicative"], d["gain_correction"], d["sa_correction"],
        d["equivalent_beam_angle"], d["frequency_nominal"], d["transmit_duration_nominal"][:, 0].contiguous(),
        pulse_length=d["pulse_length"], gain_is_table=True, sa_is_table=True)
    n_r = len(np.arange(0, float((S - 1) * 2.56e-4 * float(d["sound_speed_indicative"].max()) / 2) + 1.0, 1.0)) - 1
    sets.append((d["backscatter_r"], coef, d["ping_time_ns"], n_r, torch.empty((C, P, S), dtype=dt, device="cuda")))
    del d
streams = [torch.cuda.Stream() for _ in range(2)]
t = ops.Timer()


def view(k, off):
    raw, coef, ns, n_r, sv = sets[k]
    nsv = ns[off:off + L]
    e0, _ = sharding.global_time_grid(nsv.cpu().numpy(), 20_000_000_000)
    n_t = L // 20 + 1
    bs = ops.time_bin_offsets(nsv, e0, 20_000_000_000, n_t)
    mv = torch.empty((C, n_t, n_r), dtype=dt, device="cuda")
    return raw[:, off:off + L], coef[:, off:off + L].contiguous(), bs, n_t, n_r, sv[:, off:off + L], mv


def run(va, vb, two):
    cur = torch.cuda.current_stream()
    for st, (raw, coef, bs, n_t, n_r, sv, mv) in zip(streams if two else (cur, cur), (va, vb)):
        if two:
            st.wait_stream(cur)
        with torch.cuda.stream(st):
            ops.sv_mvbs_fused(raw, coef, bs, n_t, 1.0, n_r, dtype=dt, sv_out=sv, mvbs_out=mv)
    if two:
        for st in streams:
            cur.wait_stream(st)


va = view(0, 0)
n = 2 * C * L * S
for off in (None, 0, 20, 100, 500, 1000, 5000, 20000, 100000, 200000, 0):
    vb = view(1, off or 0)
    two = off is not None
    run(va, vb, two); torch.cuda.synchronize()
    ms = []
    for _ in range(5):
        t.start(); run(va, vb, two); t.stop(); ms.append(t.elapsed_ms())
    m = float(np.median(ms))
    print(f"{'one stream' if not two else 'B starts %6d pings in' % off:28s} {m:7.3f} ms  {n * 12 / m / 1e9 / 8:.3f}", flush=True)
