"""EK80 CW complex (no pulse compression) at kernel level: 2 x 40 000 x 8192 x 4 sectors, float32 and float64 planes --
development aid."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from echopype_amd import _lib, ops

C, P, S, B = 2, 40000, 8192, 4
g = torch.Generator(device="cuda"); g.manual_seed(1)
cc = np.zeros((C, P, _lib.NCCOEF)); cc[..., _lib.CC_RA] = 8e-6; cc[..., _lib.CC_RB] = 750.0; cc[..., _lib.CC_PSCALE] = 1.0
cc[..., _lib.CC_SHIFT] = 0.19; cc[..., _lib.CC_ALPHA2] = 0.02; cc[..., _lib.CC_A] = -30.0
ccd = torch.from_numpy(cc).cuda()
t = ops.Timer()
for plane in (torch.float32, torch.float64):
    re = (torch.randn((C, P, S, B), generator=g, device="cuda", dtype=torch.float32) * 1e-3).to(plane)
    im = (torch.randn((C, P, S, B), generator=g, device="cuda", dtype=torch.float32) * 1e-3).to(plane)
    for out_dt in (torch.float64, torch.float32):
        for want_range in (False, True, "stats only"):
            if want_range == "stats only":  # what compute_Sv asks for: echo_range stays lazy, its statistics come along
                f = lambda: ops.sv_complex(re, im, ccd, dtype=out_dt, want_range=False, want_range_stats=True)
            else:
                f = lambda: ops.sv_complex(re, im, ccd, dtype=out_dt, want_range=want_range)
            f(); torch.cuda.synchronize(); ms = []
            for _ in range(3):
                t.start(); f(); t.stop(); ms.append(t.elapsed_ms())
            m = float(np.median(ms)); n = C * P * S
            bps = B * 2 * re.element_size() + (2 if want_range is True else 1) * (8 if out_dt == torch.float64 else 4)
            print(f"CW planes {str(plane)[6:]} -> {str(out_dt)[6:]}{' + echo_range' if want_range is True else ' + range statistics' if want_range else ''}: {m:7.3f} ms  {n/m/1e6:7.1f} Gsamp/s  {n*bps/m/1e9:5.2f} TB/s", flush=True)
    del re, im
