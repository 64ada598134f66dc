"""Where compute_Sv (EK80 BB complex) spends its host time, per-ping parameters on the host / in HBM (development aid)."""
import cProfile, logging, pstats, sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
import echopype_amd as ep
C, P, S, B = 2, 20000, 2048, 4
d = ep.synth.ek80_numpy(C, 4, 64, B)
g = torch.Generator(device="cuda"); g.manual_seed(1)
re = torch.randn((C, P, S, B), generator=g, device="cuda", dtype=torch.float32) * 1e-3
im = torch.randn((C, P, S, B), generator=g, device="cuda", dtype=torch.float32) * 1e-3
p = np.arange(P)
d.update(backscatter_r=ep.DeviceArray(re), backscatter_i=ep.DeviceArray(im), sample_interval=np.full((C, P), 8e-6),
         sound_speed=np.tile(1500.0 + 0.5 * np.sin(2 * np.pi * p / 1e5), (C, 1)),
         ping_time=np.datetime64("2026-05-01T00:00:00", "ns") + (p * 1_000_000_000).astype("timedelta64[ns]"))
ed = ep.echodata.from_ek80_arrays(d, ep.synth.ek80_filters())
logging.disable(logging.WARNING)
f = lambda: ep.calibrate.compute_Sv(ed, waveform_mode="BB", encode_mode="complex")
for resident in (False, True):
    if resident:
        ed.to_device()
    r = f(); torch.cuda.synchronize(); ts = []
    for _ in range(5):
        del r; torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print(f"parameters {'in HBM' if resident else 'on the host'}: compute_Sv {np.median(ts)*1e3:7.2f} ms")
    pr = cProfile.Profile(); pr.enable(); r = f(); torch.cuda.synchronize(); pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(30)
