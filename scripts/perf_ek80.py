"""Timing probe of the EK80 complex kernel (development aid)."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from echopype_amd import _lib, ops
C, P, S, B = (int(x) for x in (sys.argv[1:5] if len(sys.argv) > 4 else (2, 20000, 8192, 4)))
taps = int(sys.argv[5]) if len(sys.argv) > 5 else 177
g = torch.Generator(device="cuda"); g.manual_seed(1)
n = C * P * S
cc = np.zeros((C, P, _lib.NCCOEF)); cc[..., _lib.CC_RA] = 8e-6; cc[..., _lib.CC_RB] = 750.0; cc[..., _lib.CC_PSCALE] = 1.0
cc[..., _lib.CC_SHIFT] = 0.19; cc[..., _lib.CC_ALPHA2] = 0.02; cc[..., _lib.CC_A] = -30.0
ccd = torch.from_numpy(cc).cuda()
rep = (torch.randn(2 * C * taps, generator=g, device="cuda", dtype=torch.float32) * 0.1).contiguous()
off = torch.arange(0, (C + 1) * taps, taps, dtype=torch.int32, device="cuda")
t = ops.Timer()
for in_dt in (torch.float64, torch.float32):
    re = (torch.randn((C, P, S, B), generator=g, device="cuda", dtype=torch.float32) * 1e-3).to(in_dt)
    im = (torch.randn((C, P, S, B), generator=g, device="cuda", dtype=torch.float32) * 1e-3).to(in_dt)
    esz = 8 if in_dt == torch.float64 else 4
    for out_dt in (torch.float64, torch.float32):
        for name, kw in (("BB-fft", dict(replica=rep, replica_off=off, max_taps=taps, method="fft")),
                         ("BB-direct", dict(replica=rep, replica_off=off, max_taps=taps, method="direct")), ("CW", dict())):
            fn = lambda: ops.sv_complex(re, im, ccd, dtype=out_dt, want_range=False, **kw)
            fn(); torch.cuda.synchronize(); ms = []
            for _ in range(5):
                t.start(); fn(); t.stop(); ms.append(t.elapsed_ms())
            m = float(np.median(ms))
            osz = 8 if out_dt == torch.float64 else 4
            flops = 8.0 * taps * n if name.startswith("BB") else 0
            print(f"{name:9s} in={str(in_dt)[6:]:8s} out/acc={str(out_dt)[6:]:8s} {m:8.3f} ms {n/m/1e6:7.1f} Gsamp/s "
                  f"{n*(2*B*esz+osz)/m/1e9:5.2f} TB/s {flops/m/1e9:6.1f} TFLOP/s", flush=True)
    del re, im
