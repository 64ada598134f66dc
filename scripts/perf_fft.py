"""Timing probe of the LDS-FFT pulse-compression kernel (development aid): cfg4-shaped planes at a reduced ping count,
every library variant named on the command line in its own process (the library is chosen at import).
    python scripts/perf_fft.py [P] lib1.so lib2.so ..."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.environ.get("PERF_FFT_CHILD"):
    import numpy as np
    import torch

    sys.path.insert(0, ROOT)
    from echopype_amd import _lib, ops

    P = int(os.environ["PERF_FFT_P"])
    C, S, B, taps = 2, 8192, 4, 177
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    cc = np.zeros((C, P, _lib.NCCOEF)); cc[..., _lib.CC_RA] = 8e-6; cc[..., _lib.CC_RB] = 750.0; cc[..., _lib.CC_PSCALE] = 1.0
    cc[..., _lib.CC_SHIFT] = 0.19; cc[..., _lib.CC_ALPHA2] = 0.02; cc[..., _lib.CC_A] = -30.0
    ccd = torch.from_numpy(cc).cuda()
    rep = (torch.randn(2 * C * taps, generator=g, device="cuda", dtype=torch.float32) * 0.1).contiguous()
    off = torch.arange(0, (C + 1) * taps, taps, dtype=torch.int32, device="cuda")
    t = ops.Timer()
    res = {}
    for in_dt in ([torch.float32] if not os.environ.get("PERF_FFT_P64") else [torch.float32, torch.float64]):
        re = (torch.randn((C, P, S, B), generator=g, device="cuda", dtype=torch.float32) * 1e-3).to(in_dt)
        im = (torch.randn((C, P, S, B), generator=g, device="cuda", dtype=torch.float32) * 1e-3).to(in_dt)
        for out_dt in (torch.float64, torch.float32):
            fn = lambda: ops.sv_complex(re, im, ccd, dtype=out_dt, want_range=False, replica=rep, replica_off=off,  # noqa: E731
                                        max_taps=taps, method="fft")
            fn(); torch.cuda.synchronize(); ms = []
            for _ in range(7):
                t.start(); fn(); t.stop(); ms.append(t.elapsed_ms())
            res[f"{str(in_dt)[11:]}->{str(out_dt)[11:]}"] = round(float(np.median(ms)) * 200000 / P, 2)
        del re, im
    print(json.dumps(res))
    sys.exit(0)
args = sys.argv[1:]
P = 40000
if args and args[0].isdigit():
    P = int(args.pop(0))
for rnd in range(int(os.environ.get("ROUNDS", "2"))):
    for lib in args:
        env = dict(os.environ, PERF_FFT_CHILD="1", PERF_FFT_P=str(P), ECHOPYPE_AMD_LIB=os.path.abspath(lib))
        r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True)
        print(os.path.basename(lib), "(ms per 2x200000x8192 volume)", r.stdout.strip() or r.stderr[-400:], flush=True)
