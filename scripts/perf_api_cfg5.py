"""API-level cost of the metric's own tiles: compute_Sv(ed_tile) -> compute_MVBS(ds) per resident tile of
4 x 250 000 x 4096 -- with the results dropped unread, read one tile late (what bench.py does), read at once -- and the
host synchronisations torch sees (torch.cuda.set_sync_debug_mode).  Development aid for the pipelined API route.
    python scripts/perf_api_cfg5.py [n_tiles] [host-profile]"""
import collections, cProfile, logging, pstats, sys, time, warnings
import numpy as np, torch
sys.path.insert(0, ".")
import echopype_amd as ep

C, P, S = 4, 250_000, 4096
NT = int(sys.argv[1]) if len(sys.argv) > 1 else 4
logging.disable(logging.WARNING)
eds = []
for i in range(NT):
    dd = ep.synth.ek60_device(C, P, S, seed=20260505 + i, ping0=i * P, ss_every=1)
    d = ep.synth.ek60_numpy(C, 4, 8)
    h = ep.synth.ek60_params(C, P, ping0=i * P, ss_every=1)
    for k in ("sample_interval", "transmit_duration_nominal", "transmit_power", "sound_speed_indicative",
              "absorption_indicative"):
        d[k] = h[k]
    d["ping_time"] = h["ping_time"] + np.timedelta64(10, "s")
    d["backscatter_r"] = ep.DeviceArray(dd["backscatter_r"])
    eds.append(ep.echodata.from_ek60_arrays(d).to_device())


def api_pass(lag):
    pending = collections.deque()
    for ed in eds:
        ds = ep.calibrate.compute_Sv(ed)
        mv = ep.commongrid.compute_MVBS(ds, range_bin="1m", ping_time_bin="20s")
        if lag is None:
            continue
        pending.append((ds, mv))
        while len(pending) > lag:
            pending.popleft()[1]["Sv"].shape
    while pending:
        pending.popleft()[1]["Sv"].shape


def timeit(f, n=5):
    f(); f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for name, lag in (("results dropped unread", None), ("read one tile late", 1), ("read two tiles late", 2), ("read at once", 0)):
    ms = timeit(lambda: api_pass(lag))
    print(f"{name:24s}: {ms:7.2f} ms per pass of {NT} tiles = {ms / NT:6.2f} ms per tile = "
          f"{C * P * S * 12 / (ms / NT * 1e-3) / 8e12:.3f} of 8 TB/s", flush=True)

if len(sys.argv) > 2:
    pr = cProfile.Profile(); pr.enable()
    for _ in range(5):
        api_pass(1)
    torch.cuda.synchronize(); pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(28)
torch.cuda.set_sync_debug_mode("warn")
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    api_pass(None)
torch.cuda.set_sync_debug_mode("default")
seen = collections.Counter(f"{x.filename.split('/')[-1]}:{x.lineno}" for x in w if "synchroniz" in str(x.message))
print("host synchronisations in one pass with the results unread:", dict(seen) or "none")
