"""API-level cost of the metric's own tiles: compute_Sv(ed_tile) -> compute_MVBS(ds) per resident tile of
4 x 250 000 x 4096, against the ops-level harness on the same tiles; lists the host synchronisations torch sees
(torch.cuda.set_sync_debug_mode) -- development aid for the pipelined API route."""
import logging, sys, time, warnings
import numpy as np, torch
sys.path.insert(0, ".")
import echopype_amd as ep
from echopype_amd import ops

C, P, S = 4, 250_000, 4096
NT = int(sys.argv[1]) if len(sys.argv) > 1 else 3
logging.disable(logging.WARNING)
tiles, eds = [], []
for i in range(NT):
    dd = ep.synth.ek60_device(C, P, S, seed=20260505 + i, ping0=i * P, ss_every=1)
    d = ep.synth.ek60_numpy(C, 4, 8)
    h = ep.synth.ek60_params(C, P, ping0=i * P, ss_every=1)
    for k in ("sample_interval", "transmit_duration_nominal", "transmit_power", "sound_speed_indicative",
              "absorption_indicative", "ping_time"):
        d[k] = h[k]
    d["backscatter_r"] = ep.DeviceArray(dd["backscatter_r"])
    eds.append(ep.echodata.from_ek60_arrays(d).to_device())
    tiles.append(dd)
sv_buf = torch.empty((C, P, S), dtype=torch.float64, device="cuda")

def api_pass(keep):
    out = []
    for ed in eds:
        ds = ep.calibrate.compute_Sv(ed)
        mv = ep.commongrid.compute_MVBS(ds, range_bin="1m", ping_time_bin="20s")
        out.append((ds, mv))
        if not keep:
            out.clear()
    return out

def timeit(f, n=4):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

print(f"api two calls per tile, {NT} tiles: {timeit(lambda: api_pass(False)):.2f} ms per pass "
      f"({timeit(lambda: api_pass(False)) / NT:.2f} per tile)")
mode = sys.argv[2] if len(sys.argv) > 2 else "warn"
torch.cuda.set_sync_debug_mode(mode)
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    api_pass(False)
torch.cuda.set_sync_debug_mode("default")
import collections, traceback
seen = collections.Counter(str(x.message)[:90] + " @ " + f"{x.filename.split('/')[-1]}:{x.lineno}" for x in w)
for k, v in seen.items():
    print(v, k)
