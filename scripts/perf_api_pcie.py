"""End-to-end (PCIe-inclusive) rate of the Dataset API on host-resident arrays (development aid)."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
import echopype_amd as ep
C, P, S = 4, 20000, 2000
d = ep.synth.ek60_numpy(C, P, S)
ed = ep.echodata.from_ek60_arrays(d)
n = C * P * S
import logging; logging.disable(logging.WARNING)
def run():
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ds = ep.calibrate.compute_Sv(ed)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    mv = ep.commongrid.compute_MVBS(ds, range_bin="1m", ping_time_bin="20s")
    torch.cuda.synchronize(); t2 = time.perf_counter()
    sv_host = ds["Sv"].values; mv_host = mv["Sv"].values
    t3 = time.perf_counter()
    return t1 - t0, t2 - t1, t3 - t2
run()
ts = np.array([run() for _ in range(3)]).min(axis=0)
print(f"compute_Sv (H2D 4 B/sample + kernel): {ts[0]*1e3:.1f} ms = {n/ts[0]/1e9:.2f} Gsamp/s")
print(f"compute_MVBS (device-resident Sv):     {ts[1]*1e3:.1f} ms = {n/ts[1]/1e9:.2f} Gsamp/s")
print(f".values (D2H 8 B/sample Sv):           {ts[2]*1e3:.1f} ms = {n*8/ts[2]/1e9:.2f} GB/s")
def fused():
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ds, mv = ep.compute_Sv_MVBS(ed, range_bin="1m", ping_time_bin="20s")
    h = mv["Sv"].values
    torch.cuda.synchronize(); return time.perf_counter() - t0
fused(); tf = min(fused() for _ in range(3))
print(f"compute_Sv_MVBS, MVBS to host, Sv left in HBM: {tf*1e3:.1f} ms = {n/tf/1e9:.2f} Gsamp/s end to end")
