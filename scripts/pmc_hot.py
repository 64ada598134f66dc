"""The VALU-bound kernels alone, a few launches each, for rocprofv3 --pmc passes (development aid):
chain pass 1 / pass 2 (fp64, 4 x 100 000 x 2000) and the four LDS-FFT pulse-compression variants (2 x 5000 x 8192 x 4)."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from echopype_amd import _lib, ops, synth

which = sys.argv[1] if len(sys.argv) > 1 else "all"
if which in ("all", "chain"):
    C, P, S = 4, 100_000, 2000
    d = synth.ek60_device(C, P, S, ss_every=1)  # a new sound speed at every ping: pass 2 = the drift kernel
    coef = ops.power_coef_ek(d["sample_interval"], d["transmit_duration_nominal"], d["transmit_power"],
                             d["sound_speed_indicative"], d["absorption_indicative"], d["gain_correction"], d["sa_correction"],
                             d["equivalent_beam_angle"], d["frequency_nominal"], d["transmit_duration_nominal"][:, 0].contiguous(),
                             pulse_length=d["pulse_length"], gain_is_table=True, sa_is_table=True)
    a2 = coef[..., 4].contiguous()
    n_t = P // 20
    bs = ops.time_bin_offsets(d["ping_time_ns"], int(d["ping_time_ns"][0].item()), 20_000_000_000, n_t)
    n_r = len(np.arange(0, float((S - 1) * 2.56e-4 * 1500.5 / 2) + 1.0, 1.0)) - 1
    for _ in range(3):
        sv, _, nz = ops.sv_noise_fused(d["backscatter_r"], coef, a2, 20, 50)
        res = ops.sv_denoise_mvbs(d["backscatter_r"], coef, a2, nz, 20, 3.0, bs, n_t, 1.0, n_r, want_noise=True)
        res = ops.sv_mvbs_fused(d["backscatter_r"], coef, bs, n_t, 1.0, n_r)
        ops.mvbs(res["Sv"], bs, n_t, 1.0, n_r, coef=coef, coef_as_stored=True)
    torch.cuda.synchronize()
    print("chain samples per launch", C * P * S)
    del d, sv, res
if which in ("all", "fft"):
    C, P, S, B, taps = 2, 5000, 8192, 4, 177
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    cc = np.zeros((C, P, _lib.NCCOEF)); cc[..., _lib.CC_RA] = 8e-6; cc[..., _lib.CC_RB] = 750.0; cc[..., _lib.CC_PSCALE] = 1.0
    cc[..., _lib.CC_SHIFT] = 0.19; cc[..., _lib.CC_ALPHA2] = 0.02; cc[..., _lib.CC_A] = -30.0
    ccd = torch.from_numpy(cc).cuda()
    rep = (torch.randn(2 * C * taps, generator=g, device="cuda", dtype=torch.float32) * 0.1).contiguous()
    off = torch.arange(0, (C + 1) * taps, taps, dtype=torch.int32, device="cuda")
    re = torch.randn((C, P, S, B), generator=g, device="cuda", dtype=torch.float32) * 1e-3
    im = torch.randn((C, P, S, B), generator=g, device="cuda", dtype=torch.float32) * 1e-3
    for out_dt in (torch.float64, torch.float32):
        for _ in range(3):
            ops.sv_complex(re, im, ccd, replica=rep, replica_off=off, max_taps=taps, dtype=out_dt, want_range=False, method="fft")
    torch.cuda.synchronize()
    print("fft samples per launch", C * P * S)
