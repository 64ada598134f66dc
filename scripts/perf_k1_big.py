"""K1 / fused at the bench volume (development aid): is the fused kernel at the streaming ceiling?"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from echopype_amd import ops, synth
C, P, S = 4, 500000, 2000
d = synth.ek60_device(C, P, S)
cf = ops.power_coef_ek(d["sample_interval"], d["transmit_duration_nominal"], d["transmit_power"],
    d["sound_speed_indicative"], d["absorption_indicative"], d["gain_correction"], d["sa_correction"],
    d["equivalent_beam_angle"], d["frequency_nominal"], d["transmit_duration_nominal"][:, 0].contiguous(),
    pulse_length=d["pulse_length"], gain_is_table=True, sa_is_table=True)
n = C * P * S
out = torch.empty((C, P, S), dtype=torch.float64, device="cuda")
t = ops.Timer()
def timeit(name, fn, nbytes):
    fn(); torch.cuda.synchronize(); ms = []
    for _ in range(7):
        t.start(); fn(); t.stop(); ms.append(t.elapsed_ms())
    m = float(np.median(ms)); print(f"{name:46s} {m:8.3f} ms {n/m/1e6:7.1f} Gsamp/s {nbytes/m/1e9:6.2f} TB/s", flush=True)
timeit("K1 sv_power f64 (no range)", lambda: ops.sv_power(d["backscatter_r"], cf, want_range=False, out=out), n * 12)
timeit("torch f32->f64 cast", lambda: out.copy_(d["backscatter_r"]), n * 12)
ns = d["ping_time_ns"]; t0 = int(ns[0].item()); n_t = P // 20
bs = ops.time_bin_offsets(ns, t0, 20_000_000_000, n_t)
mv = torch.empty((C, n_t, 385), dtype=torch.float64, device="cuda")
timeit("fused f64 Sv+MVBS", lambda: ops.sv_mvbs_fused(d["backscatter_r"], cf, bs, n_t, 1.0, 385, sv_out=out, mvbs_out=mv), n * 12)
timeit("fused f64 MVBS only", lambda: ops.sv_mvbs_fused(d["backscatter_r"], cf, bs, n_t, 1.0, 385, want_sv=False, mvbs_out=mv), n * 4)
timeit("torch f64 fill", lambda: out.fill_(1.0), n * 8)
timeit("torch f32 sum", lambda: d["backscatter_r"].sum(), n * 4)
