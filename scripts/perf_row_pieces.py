"""A/B: echo_range / depth written from the coefficient rows by workgroups striding over the rows (EPA_ROW_PIECES=0)
against one-piece workgroups (round 6).  4 x P x 2000, HIP events.  Run twice: EPA_ROW_PIECES=0 python ... / python ..."""
import os, sys
sys.path.insert(0, ".")
import numpy as np, torch
import echopype_amd as ep
from echopype_amd import ops
C, S = 4, 2000
P = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000
d = ep.synth.ek60_device(C, P, S, ss_every=1)
coef = ops.power_coef_ek(d["sample_interval"], d["transmit_duration_nominal"], d["transmit_power"],
    d["sound_speed_indicative"], d["absorption_indicative"], d["gain_correction"], d["sa_correction"],
    d["equivalent_beam_angle"], d["frequency_nominal"], d["transmit_duration_nominal"][:, 0].contiguous(),
    pulse_length=d["pulse_length"], gain_is_table=True, sa_is_table=True)
raw = d["backscatter_r"]
scale = torch.full((C, P), 0.97, dtype=torch.float64, device="cuda"); offset = torch.full((C, P), 5.0, dtype=torch.float64, device="cuda")
n = C * P * S
tm = ops.Timer()
def timeit(name, fn, bps, reps=8):
    fn(); fn(); torch.cuda.synchronize(); ms = []
    for _ in range(reps):
        tm.start(); fn(); tm.stop(); ms.append(tm.elapsed_ms())
    m = float(np.median(ms))
    print(f"EPA_ROW_PIECES={os.environ.get('EPA_ROW_PIECES', '1')} {name:46s} {m:8.3f} ms  {n * bps / m / 1e9:5.2f} TB/s  frac {n * bps / m / 1e9 / 8:.3f}", flush=True)
for dt, b in ((torch.float64, 8), (torch.float32, 4)):
    tag = str(dt)[6:]
    timeit(f"{tag} range_power (masked)", lambda: ops.range_power(raw, coef, dtype=dt), 4 + b)
    timeit(f"{tag} depth_rows from rows, no statistics", lambda: ops.depth_rows(scale, offset, coef=coef, mask_raw=raw, shape=(C, P, S), dtype=dt, want_stats=False), 4 + b)
    timeit(f"{tag} depth_rows from rows + statistics", lambda: ops.depth_rows(scale, offset, coef=coef, mask_raw=raw, shape=(C, P, S), dtype=dt), 4 + b)
for dt, b in ((torch.float64, 8), (torch.float32, 4)):
    tag = str(dt)[6:]
    er = ops.range_power(raw, coef, dtype=dt)
    timeit(f"{tag} depth_rows of an array + statistics", lambda: ops.depth_rows(scale, offset, range=er), 2 * b)
    del er
