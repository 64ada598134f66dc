"""The four (output, transform) precisions of the LDS-FFT pulse compression on 2 x 20 000 x 8192 x 4 float32 planes --
development aid."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from echopype_amd import _lib, ops

C, P, S, B, taps = 2, 20000, 8192, 4, 177
g = torch.Generator(device="cuda"); g.manual_seed(1)
cc = np.zeros((C, P, _lib.NCCOEF)); cc[..., _lib.CC_RA] = 8e-6; cc[..., _lib.CC_RB] = 750.0 + 0.25 * np.sin(np.arange(P) / 3e3); cc[..., _lib.CC_PSCALE] = 1.0
cc[..., _lib.CC_SHIFT] = 0.19; cc[..., _lib.CC_ALPHA2] = 0.02; cc[..., _lib.CC_A] = -30.0
ccd = torch.from_numpy(cc).cuda()
rep = (torch.randn(2 * C * taps, generator=g, device="cuda", dtype=torch.float32) * 0.1).contiguous()
off = torch.arange(0, (C + 1) * taps, taps, dtype=torch.int32, device="cuda")
re = torch.randn((C, P, S, B), generator=g, device="cuda", dtype=torch.float32) * 1e-3
im = torch.randn((C, P, S, B), generator=g, device="cuda", dtype=torch.float32) * 1e-3
t = ops.Timer()
n = C * P * S
ref = None
for out_dt in (torch.float64, torch.float32):
    for fft_dt in ("float64", "float32"):
        f = lambda: ops.sv_complex(re, im, ccd, replica=rep, replica_off=off, max_taps=taps, dtype=out_dt, want_range=False,
                                   method="fft", fft_dtype=fft_dt)
        r = f(); torch.cuda.synchronize(); ms = []
        for _ in range(4):
            t.start(); f(); t.stop(); ms.append(t.elapsed_ms())
        m = float(np.median(ms))
        sv = r["out"].double()
        if ref is None:
            ref = sv
        d = (sv - ref).abs()
        ok = ~torch.isnan(d)
        bps = 32 + (8 if out_dt == torch.float64 else 4)
        print(f"Sv {str(out_dt)[6:]:8s} transform complex{'128' if fft_dt == 'float64' else '64 '}: {m:7.3f} ms  {n/m/1e6:6.1f} Gsamp/s  {n*bps/m/1e9:5.2f} TB/s   "
              f"max |dSv| vs f64/complex128 {float(d[ok].max()):.2e} dB, median {float(d[ok].median()):.2e}", flush=True)
