"""Latency of the Dataset API on a small file (BASELINE configs[0] shape, EK60 2 x 10 000 x 1000): host overhead vs
kernels -- development aid."""
import cProfile, logging, pstats, sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
import echopype_amd as ep
C, P, S = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (2, 10000, 1000)))
d = ep.synth.ek60_numpy(C, P, S)
logging.disable(logging.WARNING)
for resident in (False, True):
    ed = ep.echodata.from_ek60_arrays(d)
    if resident:
        ed.to_device()
    def med(f, n=20):
        r = f(); torch.cuda.synchronize(); ts = []
        for _ in range(n):
            torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        return float(np.median(ts)) * 1e3, r
    a, ds = med(lambda: ep.calibrate.compute_Sv(ed))
    b, mv = med(lambda: ep.commongrid.compute_MVBS(ds, range_bin="1m", ping_time_bin="20s"))
    c, _ = med(lambda: ep.compute_Sv_MVBS(ed, range_bin="1m", ping_time_bin="20s"))
    e, _ = med(lambda: ep.clean.remove_background_noise(ds, 20, 50))
    print(f"{'resident' if resident else 'host arrays'}: compute_Sv {a:.2f} ms, compute_MVBS {b:.2f} ms, compute_Sv_MVBS {c:.2f} ms, remove_background_noise {e:.2f} ms  ({C*P*S/1e6:.0f} M samples)")
pr = cProfile.Profile(); pr.enable()
for _ in range(10): ep.calibrate.compute_Sv(ed)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
