"""Development aid: the fused Sv -> MVBS kernel on one cfg5 tile (4 x 250 000 x 4096) with and without the echo_range
by-products ({nanmin, nanmax, NaN count}) the API route asks for -- alternating, HIP events."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from echopype_amd import ops, synth

C, P, S = 4, 250_000, 4096
d = synth.ek60_device(C, P, S, ss_every=1)
cf = ops.power_coef_ek(d["sample_interval"], d["transmit_duration_nominal"], d["transmit_power"],
    d["sound_speed_indicative"], d["absorption_indicative"], d["gain_correction"], d["sa_correction"],
    d["equivalent_beam_angle"], d["frequency_nominal"], d["transmit_duration_nominal"][:, 0].contiguous(),
    pulse_length=d["pulse_length"], gain_is_table=True, sa_is_table=True)
t_ns = torch.from_numpy(d["ping_time"].astype("datetime64[ns]").astype(np.int64)).cuda() if not torch.is_tensor(d["ping_time"]) else d["ping_time"]
n_t = P // 20
bs = ops.time_bin_offsets(t_ns, int(t_ns[0].item()), 20_000_000_000, n_t)
sv = torch.empty((C, P, S), dtype=torch.float64, device="cuda")
mv = torch.empty((C, n_t, 787), dtype=torch.float64, device="cuda")
t = ops.Timer()
res = {False: [], True: []}
for rep in range(12):
    for st in (False, True):
        t.start()
        ops.sv_mvbs_fused(d["backscatter_r"], cf, bs, n_t, 1.0, 787, sv_out=sv, mvbs_out=mv, want_range_max=st, want_range_stats=st)
        t.stop()
        if rep >= 2:
            res[st].append(t.elapsed_ms())
for st in (False, True):
    a = np.array(res[st])
    print("with the by-products" if st else "without             ", "median %.3f ms  min %.3f  max %.3f" % (np.median(a), a.min(), a.max()))
