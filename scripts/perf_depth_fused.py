"""A/B at ops level: the fused Sv -> MVBS kernel binned on the echo_range against the same kernel binned on depth
(epa_sv_mvbs_fused_depth), 4 x P x 2000 fp64, a new sound speed at every ping.  HIP events round each launch."""
import sys
sys.path.insert(0, ".")
import numpy as np, torch
import echopype_amd as ep
from echopype_amd import ops, _lib

C, S = 4, 2000
P = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
dt = torch.float64 if (len(sys.argv) < 3 or sys.argv[2] == "f64") else torch.float32
dd = ep.synth.ek60_numpy(C, 4, 8)
h = ep.synth.ek60_params(C, P, ss_every=1)
for k in ("sample_interval", "transmit_duration_nominal", "transmit_power", "sound_speed_indicative", "absorption_indicative"):
    dd[k] = h[k]
dd["ping_time"] = h["ping_time"]
dd["backscatter_r"] = ep.DeviceArray(ep.synth.ek60_device(C, P, S, seed=20260509, ss_every=1)["backscatter_r"])
ed = ep.echodata.from_ek60_arrays(dd).to_device()
cal = ep.calibrate.api.CALIBRATOR["EK60"](ed, None, None, None, dtype="float64" if dt == torch.float64 else "float32")
raw, coef, flags, _ = cal._power_inputs("Sv")
ns = torch.from_numpy(h["ping_time"].astype("datetime64[ns]").astype(np.int64)).cuda()
e0 = int(ns[0].item()); bin_ns = 20_000_000_000
n_t = int((int(ns[-1].item()) - e0) // bin_ns) + 1
bs = ops.time_bin_offsets(ns, e0, bin_ns, n_t)
n_r = 395
sv = torch.empty((C, P, S), dtype=dt, device="cuda")
mv = torch.empty((C, n_t, n_r), dtype=dt, device="cuda")
one = torch.ones((C, P), dtype=torch.float64, device="cuda")
zero = torch.zeros((C, P), dtype=torch.float64, device="cuda")
five = torch.full((C, P), 5.0, dtype=torch.float64, device="cuda")
neg = -one
off400 = torch.full((C, P), 390.0, dtype=torch.float64, device="cuda")

def t(f, n=12):
    for _ in range(3): f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return np.median(ts), min(ts)

n = C * P * S
def rep(name, f, bps):
    med, mn = t(f)
    print(f"{name:46s} {med:7.3f} ms (min {mn:.3f})  {n * bps / med / 1e9:6.2f} TB/s... frac {n * bps / med / 1e9 / 8:.3f}", flush=True)

bps = 12 if dt == torch.float64 else 8
rep("echo_range, no stats", lambda: ops.sv_mvbs_fused(raw, coef, bs, n_t, 1.0, n_r, dtype=dt, sv_out=sv, mvbs_out=mv), bps)
rep("echo_range, stats", lambda: ops.sv_mvbs_fused(raw, coef, bs, n_t, 1.0, n_r, dtype=dt, sv_out=sv, mvbs_out=mv, want_range_stats=True), bps)
rep("depth scale 1 offset 0", lambda: ops.sv_mvbs_fused_depth(raw, coef, one, zero, bs, n_t, 1.0, n_r, dtype=dt, sv_out=sv, mvbs_out=mv), bps)
rep("depth scale 1 offset 5", lambda: ops.sv_mvbs_fused_depth(raw, coef, one, five, bs, n_t, 1.0, n_r, dtype=dt, sv_out=sv, mvbs_out=mv), bps)
rep("depth scale -1 offset 390", lambda: ops.sv_mvbs_fused_depth(raw, coef, neg, off400, bs, n_t, 1.0, n_r, dtype=dt, sv_out=sv, mvbs_out=mv), bps)
rep("depth, bins only", lambda: ops.sv_mvbs_fused_depth(raw, coef, one, five, bs, n_t, 1.0, n_r, dtype=dt, want_sv=False, mvbs_out=mv), bps - (8 if dt == torch.float64 else 4))
rep("depth written too", lambda: ops.sv_mvbs_fused_depth(raw, coef, one, five, bs, n_t, 1.0, n_r, dtype=dt, sv_out=sv, mvbs_out=mv, want_depth=True), bps + (8 if dt == torch.float64 else 4))
rep("echo_range, stats (again)", lambda: ops.sv_mvbs_fused(raw, coef, bs, n_t, 1.0, n_r, dtype=dt, sv_out=sv, mvbs_out=mv, want_range_stats=True), bps)
rep("depth scale 1 offset 5 (again)", lambda: ops.sv_mvbs_fused_depth(raw, coef, one, five, bs, n_t, 1.0, n_r, dtype=dt, sv_out=sv, mvbs_out=mv), bps)
rep("echo_range, no stats (again)", lambda: ops.sv_mvbs_fused(raw, coef, bs, n_t, 1.0, n_r, dtype=dt, sv_out=sv, mvbs_out=mv), bps)
