"""Host wall time of the reference's two calls on a resident 4 x 250 000 x 4096 tile (the headline's tile), call by call and
by internal time (cProfile, 200 calls so that the profiler's own cost is spread thin) -- development aid, round 6."""
import cProfile, logging, pstats, sys, time, io
import numpy as np, torch
sys.path.insert(0, ".")
import echopype_amd as ep
C, P, S = 4, 50000, 4096   # (the host work depends on C and P only through O(P) host passes)
P = int(sys.argv[1]) if len(sys.argv) > 1 else P
dd = ep.synth.ek60_device(C, P, S)
d = ep.synth.ek60_numpy(C, 4, 8)
h = ep.synth.ek60_params(C, P, ss_every=1)
for k in ("sample_interval", "transmit_duration_nominal", "transmit_power", "sound_speed_indicative", "absorption_indicative"):
    d[k] = h[k]
d["ping_time"] = h["ping_time"]
d["backscatter_r"] = ep.DeviceArray(dd["backscatter_r"])
logging.disable(logging.WARNING)
ed = ep.echodata.from_ek60_arrays(d).to_device()
def two():
    ds = ep.calibrate.compute_Sv(ed)
    return ds, ep.commongrid.compute_MVBS(ds, range_bin="1m", ping_time_bin="20s")
prev = None
for _ in range(5):
    cur = two()
    if prev: prev[1]["Sv"].shape
    prev = cur
torch.cuda.synchronize()
N = 200
t_sv = t_mv = t_rd = 0.0
prev = None
for _ in range(N):
    t0 = time.perf_counter(); ds = ep.calibrate.compute_Sv(ed); t1 = time.perf_counter()
    mv = ep.commongrid.compute_MVBS(ds, range_bin="1m", ping_time_bin="20s"); t2 = time.perf_counter()
    t_sv += t1 - t0; t_mv += t2 - t1
    torch.cuda.synchronize()   # (the read below then waits for nothing: pure assembly cost)
    t3 = time.perf_counter(); mv["Sv"].shape; t_rd += time.perf_counter() - t3
print(f"P = {P}: compute_Sv {t_sv / N * 1e3:.3f} ms, compute_MVBS {t_mv / N * 1e3:.3f} ms, assembly of the deferred dataset {t_rd / N * 1e3:.3f} ms (host, per call)")
pr = cProfile.Profile(); pr.enable()
for _ in range(N):
    ds, mv = two(); torch.cuda.synchronize(); mv["Sv"].shape
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(40); print(s.getvalue()[:7000])
