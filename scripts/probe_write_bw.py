"""Development aid: what the HBM does for a pure write stream / a pure read / a 1:2 read:write copy of the fused kernel's
volume, next to the fused kernel with and without its Sv store (4 x 500 000 x 2000)."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from echopype_amd import ops, synth

C, P, S = 4, 500_000, 2000
n = C * P * S
t = ops.Timer()
def timeit(name, fn, nbytes, reps=5):
    fn(); torch.cuda.synchronize(); ms = []
    for _ in range(reps):
        t.start(); fn(); t.stop(); ms.append(t.elapsed_ms())
    m = float(np.median(ms)); print(f"{name:58s} {m:8.3f} ms  {nbytes / m / 1e9:6.2f} TB/s", flush=True)
a = torch.empty(n, dtype=torch.float64, device="cuda")
b = torch.empty(n, dtype=torch.float32, device="cuda").normal_()
timeit("fill of 8 B/sample (write only)", lambda: a.fill_(1.0), n * 8)
timeit("sum of the f32 samples (read only, 4 B/sample)", lambda: b.sum(), n * 4)
timeit("f32 -> f64 conversion copy (4 B read + 8 B written)", lambda: a.copy_(b), n * 12)
del a, b
d = synth.ek60_device(C, P, S, ss_every=1)
cf = ops.power_coef_ek(d["sample_interval"], d["transmit_duration_nominal"], d["transmit_power"],
    d["sound_speed_indicative"], d["absorption_indicative"], d["gain_correction"], d["sa_correction"],
    d["equivalent_beam_angle"], d["frequency_nominal"], d["transmit_duration_nominal"][:, 0].contiguous(),
    pulse_length=d["pulse_length"], gain_is_table=True, sa_is_table=True)
pt = d["ping_time"]
t_ns = pt if torch.is_tensor(pt) else torch.from_numpy(np.asarray(pt).astype("datetime64[ns]").astype(np.int64)).cuda()
n_t = P // 20
bs = ops.time_bin_offsets(t_ns, int(t_ns[0].item()), 20_000_000_000, n_t)
sv = torch.empty((C, P, S), dtype=torch.float64, device="cuda")
mv = torch.empty((C, n_t, 384), dtype=torch.float64, device="cuda")
timeit("fused kernel, Sv written (12 B/sample)", lambda: ops.sv_mvbs_fused(d["backscatter_r"], cf, bs, n_t, 1.0, 384, sv_out=sv, mvbs_out=mv), n * 12)
timeit("fused kernel, bins only (4 B/sample)", lambda: ops.sv_mvbs_fused(d["backscatter_r"], cf, bs, n_t, 1.0, 384, want_sv=False, mvbs_out=mv), n * 4)
