"""Development aid: EK80 BB complex through the reference's two calls (compute_Sv, compute_MVBS), file after file,
on half the cfg4 volume (2 x 100 000 x 8192 x 4 sectors), against the host syncs torch sees."""
import collections, logging, sys, time, warnings
import numpy as np
import torch
sys.path.insert(0, ".")
import echopype_amd as ep

C, P, S, B = 2, 100_000, 8192, 4
d = ep.synth.ek80_numpy(C, 4, 64, B)
g = torch.Generator(device="cuda"); g.manual_seed(1)
re = torch.randn((C, P, S, B), generator=g, device="cuda", dtype=torch.float32) * 1e-3
im = torch.randn((C, P, S, B), generator=g, device="cuda", dtype=torch.float32) * 1e-3
p = np.arange(P)
d.update(backscatter_r=ep.DeviceArray(re), backscatter_i=ep.DeviceArray(im), sample_interval=np.full((C, P), 8e-6),
         sound_speed=np.tile(1500.0 + 0.5 * np.sin(2 * np.pi * p / 1e5), (C, 1)),
         ping_time=np.datetime64("2026-05-01T00:00:00", "ns") + (p * 1_000_000_000).astype("timedelta64[ns]"))
ed = ep.echodata.from_ek80_arrays(d, ep.synth.ek80_filters()).to_device()
logging.disable(logging.WARNING)


def one():
    ds = ep.calibrate.compute_Sv(ed, waveform_mode="BB", encode_mode="complex")
    return ds, ep.commongrid.compute_MVBS(ds, range_bin="1m", ping_time_bin="20s")


def loop(n=5):
    torch.cuda.synchronize(); t0 = time.perf_counter(); prev = None
    for _ in range(n):
        cur = one()
        if prev is not None:
            prev[1]["Sv"].shape
        prev = cur
    prev[1]["Sv"].shape
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


loop(2)
print("two calls, file after file: %.2f ms per file of %d samples" % (loop(), C * P * S))
t = ep.ops.Timer(); t.start(); r = one(); t.stop(); print("GPU time of one file's calls (HIP events): %.2f ms" % t.elapsed_ms())
torch.cuda.set_sync_debug_mode("warn")
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    one()
torch.cuda.set_sync_debug_mode("default")
seen = collections.Counter(f"{x.filename.split('/')[-1]}:{x.lineno}" for x in w if "synchroniz" in str(x.message))
print("host synchronisations in the two calls:", dict(seen) or "none")
