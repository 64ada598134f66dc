import sys, numpy as np, torch
sys.path.insert(0, ".")
from echopype_amd import _lib, ops
torch.manual_seed(0)
C, P, S, B, taps = 2, 12, 8192, 4, 177
g = torch.Generator(device="cuda"); g.manual_seed(3)
re = torch.randn((C, P, S, B), generator=g, device="cuda", dtype=torch.float32) * 1e-3
im = torch.randn((C, P, S, B), generator=g, device="cuda", dtype=torch.float32) * 1e-3
rep = torch.randn(2 * C * taps, generator=g, device="cuda", dtype=torch.float32) * 0.1
off = torch.arange(0, (C + 1) * taps, taps, dtype=torch.int32, device="cuda")
# strong echo per ping
layer = (torch.arange(P, device="cuda") * 701) % (S - 600) + 100
for c in range(C):
    r = rep.view(-1, 2)[c * taps:(c + 1) * taps]
    for p in range(P):
        for b in range(B):
            re[c, p, layer[p]:layer[p] + taps, b] += 5.0 * r[:, 0]
            im[c, p, layer[p]:layer[p] + taps, b] += 5.0 * r[:, 1]
cc = np.zeros((C, P, _lib.NCCOEF)); cc[..., 0] = 8e-6; cc[..., 1] = 750.0; cc[..., 5] = 1.0; cc[..., 2] = 0.19; cc[..., 3] = 0.02; cc[..., 4] = -30.0
ccd = torch.from_numpy(cc).cuda()
kw = dict(replica=rep, replica_off=off, max_taps=taps, want_range=False)
d = ops.sv_complex(re, im, ccd, method="direct", dtype=torch.float64, **kw)["out"].cpu().numpy()
for dt, fd in ((torch.float64, None), (torch.float32, None), (torch.float64, "float32")):
    f = ops.sv_complex(re, im, ccd, method="fft", dtype=dt, fft_dtype=fd, **kw)["out"].cpu().numpy().astype(np.float64)
    err = np.abs(f - d)
    peak = np.nanmax(d, axis=2, keepdims=True)
    w40 = np.isfinite(d) & (d > peak - 40)
    print(dt, fd, "nan mismatch", int((np.isnan(f) != np.isnan(d)).sum()), "max err within 40 dB", np.nanmax(np.where(w40, err, 0)), "median err", np.nanmedian(err))
    idx = np.unravel_index(np.nanargmax(np.where(w40, err, 0)), err.shape)
    u = idx[1] * (S + taps - 1) + idx[2]
    print("  worst at", idx, "tile", u // (2049 - taps), "pos in tile", u % (2049 - taps), "d", d[idx], "f", f[idx], "layer", [int(x) for x in layer.cpu()][idx[1]])
