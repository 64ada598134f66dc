"""Which buffer's placement decides what the fused kernel streams?  K copies of the raw samples and K output arrays, each
its own allocation; the kernel on every (raw copy, Sv array) combination (one stream, HIP events, median of 5)."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from echopype_amd import ops, synth, sharding

K = int(sys.argv[1]) if len(sys.argv) > 1 else 4
C, P, S = 4, 250000, 2000
dt = torch.float64
d = synth.ek60_device(C, P, S, seed=20260509, ss_every=1)
coef = ops.power_coef_ek(d["sample_interval"], d["transmit_duration_nominal"], d["transmit_power"],
    d["sound_speed_indicative"], d["absorption_indicative"], d["gain_correction"], d["sa_correction"],
    d["equivalent_beam_angle"], d["frequency_nominal"], d["transmit_duration_nominal"][:, 0].contiguous(),
    pulse_length=d["pulse_length"], gain_is_table=True, sa_is_table=True)
ns = d["ping_time_ns"]
e0, _ = sharding.global_time_grid(ns.cpu().numpy(), 20_000_000_000)
n_t = P // 20
bs = ops.time_bin_offsets(ns, e0, 20_000_000_000, n_t)
n_r = len(np.arange(0, float((S - 1) * 2.56e-4 * float(d["sound_speed_indicative"].max()) / 2) + 1.0, 1.0)) - 1
raws = [d["backscatter_r"]] + [d["backscatter_r"].clone() for _ in range(K - 1)]
svs = [torch.empty((C, P, S), dtype=dt, device="cuda") for _ in range(K)]
mv = torch.empty((C, n_t, n_r), dtype=dt, device="cuda")
t = ops.Timer()
n = C * P * S
print("raw at", [hex(r.data_ptr()) for r in raws]); print("Sv at ", [hex(s.data_ptr()) for s in svs])
print("rows: raw copy; columns: Sv array; fraction of 8 TB/s")
for i, raw in enumerate(raws):
    row = []
    for j, sv in enumerate(svs):
        f = lambda: ops.sv_mvbs_fused(raw, coef, bs, n_t, 1.0, n_r, dtype=dt, sv_out=sv, mvbs_out=mv)
        f(); torch.cuda.synchronize()
        ms = []
        for _ in range(5):
            t.start(); f(); t.stop(); ms.append(t.elapsed_ms())
        row.append(n * 12 / float(np.median(ms)) / 1e9 / 8)
    print(f"raw {i}: " + "  ".join(f"{x:.3f}" for x in row), flush=True)
# bins only (no Sv store) and K1 per raw copy
for i, raw in enumerate(raws):
    f = lambda: ops.sv_mvbs_fused(raw, coef, bs, n_t, 1.0, n_r, dtype=dt, want_sv=False, mvbs_out=mv)
    f(); torch.cuda.synchronize(); ms = []
    for _ in range(5):
        t.start(); f(); t.stop(); ms.append(t.elapsed_ms())
    print(f"raw {i}: bins only {float(np.median(ms)):.3f} ms", flush=True)
