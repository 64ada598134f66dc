import sys, warnings, numpy as np, torch
sys.path.insert(0, ".")
from echopype_amd import ops
taps, S, mixed, B = 1024, 3000, True, 4
in_dtype, out_dtype = "float32", "float32"
rng = np.random.default_rng(taps + S)
C, P = 2, 5
amp = 10.0 ** rng.uniform(-7, 0, (C, P, S, 1))
re = (amp * rng.standard_normal((C, P, S, B))).astype(in_dtype)
im = (amp * rng.standard_normal((C, P, S, B))).astype(in_dtype)
re[:, :, S - 37:], im[:, :, S - 37:] = np.nan, np.nan
re[1, 1], im[1, 1] = np.nan, np.nan
re[0, 0, 100:130, B - 1] = np.nan
im[0, 2, S // 2, 0] = np.nan
re[1, 0, 200:203, 0] = np.nan
lens = [taps, max(taps // 2, 1)]
rep = np.concatenate([(rng.standard_normal(n) + 1j * rng.standard_normal(n)) * np.hanning(n + 2)[1:-1] for n in lens]).astype(np.complex64)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
repf = dev(np.stack([rep.real, rep.imag], axis=1).astype(np.float32).reshape(-1))
off = dev(np.array([0, lens[0], lens[0] + lens[1]], dtype=np.int32))
cc = np.zeros((C, P, 8))
cc[..., 0], cc[..., 1], cc[..., 2], cc[..., 3], cc[..., 4], cc[..., 5] = 2.6e-5, 750.0, 0.2, 0.02, -30.0, 1e3
cc[0, 2, 1], cc[1, 2, 3] = 751.5, 0.021
cc[:, 3, 4], cc[:, 4, 5] = -31.5, 1.1e3
kw = dict(replica=repf, replica_off=off, max_taps=taps, dtype=getattr(torch, out_dtype), want_prx=True)
args = (dev(re), dev(im), dev(cc))
d = ops.sv_complex(*args, method="direct", **kw)
f = ops.sv_complex(*args, method="fft", want_range_stats=True, **kw)
d64 = ops.sv_complex(*args, method="direct", **{**kw, "dtype": torch.float64})
pd, pf, p64 = (x["prx"].cpu().numpy().astype(np.float64) for x in (d, f, d64))
bad = np.argwhere(np.isnan(pf) != np.isnan(pd))
print("mismatches", bad)
for c, p, s in bad:
    u = p * (S + taps - 1) + s
    print("at", (c, p, s), "tile", u // (2049 - taps), "pos", u % (2049 - taps), "pd", pd[c, p, s], "pf", pf[c, p, s], "p64", p64[c, p, s],
          "peak", np.nanmax(p64[c, p]), "neighbours", pd[c, p, max(0, s - 3):s + 4], pf[c, p, max(0, s - 3):s + 4])
