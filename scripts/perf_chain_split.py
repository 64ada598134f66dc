"""The two sweeps of the chain timed INSIDE the alternating loop (pass 1, pass 2, pass 1, ...): HIP events round each
launch -- is a pass slower behind the other than alone?  Development aid.   python scripts/perf_chain_split.py [P] [ss_every]"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from echopype_amd import _lib, ops, synth
C, P, S = 4, int(sys.argv[1]) if len(sys.argv) > 1 else 500000, 2000
ss = int(sys.argv[2]) if len(sys.argv) > 2 else 1
d = synth.ek60_device(C, P, S, ss_every=ss)
coef = ops.power_coef_ek(d["sample_interval"], d["transmit_duration_nominal"], d["transmit_power"],
    d["sound_speed_indicative"], d["absorption_indicative"], d["gain_correction"], d["sa_correction"],
    d["equivalent_beam_angle"], d["frequency_nominal"], d["transmit_duration_nominal"][:, 0].contiguous(),
    pulse_length=d["pulse_length"], gain_is_table=True, sa_is_table=True)
raw = d["backscatter_r"]
a2 = coef[..., _lib.CF_ALPHA2].contiguous()
ns = d["ping_time_ns"]; t0 = int(ns[0].item()); dtb = 20_000_000_000
n_t = int((int(ns[-1].item()) - t0) // dtb) + 1
bs = ops.time_bin_offsets(ns, t0, dtb, n_t)
n = C * P * S
_, _, nz, rm = ops.sv_noise_fused(raw, coef, a2, 20, 50, want_range_max=True)
n_r = len(np.arange(0, rm + 1.0, 1.0)) - 1
print(f"-- 4 x {P} x {S}, sound speed changes every {ss} ping(s)", flush=True)
for dt, b in ((torch.float64, 8), (torch.float32, 4)):
    t1, t2 = ops.Timer(), ops.Timer()
    def p1():
        return ops.sv_noise_fused(raw, coef, a2, 20, 50, dtype=dt)[2]
    def p2(nz):
        return ops.sv_denoise_mvbs(raw, coef, a2, nz, 20, 3.0, bs, n_t, 1.0, n_r, dtype=dt, want_noise=True)
    nz = p1(); p2(nz); torch.cuda.synchronize()
    m1, m2 = [], []
    for _ in range(6):
        t1.start(); nz = p1(); t1.stop()
        t2.start(); out = p2(nz); t2.stop()
        del out
        torch.cuda.synchronize()
        m1.append(t1.elapsed_ms()); m2.append(t2.elapsed_ms())
    a1, a2_ = [], []
    for _ in range(4):
        t1.start(); nz = p1(); t1.stop(); torch.cuda.synchronize(); a1.append(t1.elapsed_ms())
    for _ in range(4):
        t2.start(); out = p2(nz); t2.stop(); torch.cuda.synchronize(); a2_.append(t2.elapsed_ms()); del out
    f = lambda ms, bps: f"{np.median(ms):7.3f} ms ({n * bps / np.median(ms) / 1e9:5.2f} TB/s)"
    print(f"{dt}: alternating  pass 1 {f(m1, 4 + b)}  pass 2 {f(m2, 4 + 2 * b)}   |   alone  pass 1 {f(a1, 4 + b)}  pass 2 {f(a2_, 4 + 2 * b)}", flush=True)
