"""Full-size parity of the other BASELINE configs, through size-independent windows:
  * configs[2] -- the two-pass chain compute_Sv -> remove_background_noise -> compute_MVBS at 4 x 500 000 x 2000: random
    ping windows (whole noise blocks / time bins) of Sv_noise, Sv_corrected and the MVBS against the oracle chain,
    fp32 vs fp64 on the corrected output;
  * configs[3] -- EK80 broadband at 2 x 200 000 x 8192 x 4 sectors (float32 planes, 105 GB): random ping windows against
    oracle.ek80 (scipy.signal.convolve per ping and sector);
  * configs[4]'s range depth -- epa_sv_mvbs_fused at 4 x N x 4096 with its 787 one-metre range bins against the oracle.
Each module-scoped volume is freed before the next one is built."""
import gc

import numpy as np
import pytest

from bb_tolerance import assert_bb_close
from oracle import calibrate as ocal
from oracle import clean as oclean
from oracle import commongrid as ogrid

pytestmark = pytest.mark.gpu
BIN_NS = 20_000_000_000


def _need(gb):
    import torch

    if not torch.cuda.is_available():
        pytest.fail("needs a GPU")
    gc.collect()
    torch.cuda.empty_cache()
    free, _ = torch.cuda.mem_get_info()
    if free < gb * 2**30:
        pytest.skip(f"needs ~{gb} GB of free HBM")
    return torch


def _coef(ops, d, tau0):
    return ops.power_coef_ek(
        d["sample_interval"], d["transmit_duration_nominal"], d["transmit_power"], d["sound_speed_indicative"],
        d["absorption_indicative"], d["gain_correction"], d["sa_correction"], d["equivalent_beam_angle"],
        d["frequency_nominal"], tau0, pulse_length=d["pulse_length"], gain_is_table=True, sa_is_table=True)


def _oracle_window(d, p0, w):
    h = {k: d[k][:, p0:p0 + w].cpu().numpy() for k in ("sample_interval", "transmit_duration_nominal", "transmit_power",
                                                       "sound_speed_indicative", "absorption_indicative", "backscatter_r")}
    for k in ("gain_correction", "sa_correction", "pulse_length", "equivalent_beam_angle", "frequency_nominal"):
        h[k] = d[k].cpu().numpy()
    gain = ocal.vend_cal_params_power(h["transmit_duration_nominal"], h["pulse_length"], h["gain_correction"])
    sa = ocal.vend_cal_params_power(h["transmit_duration_nominal"], h["pulse_length"], h["sa_correction"])
    sv, er = ocal.cal_power_ek(
        h["backscatter_r"], sonar="EK60", cal_type="Sv", sample_interval=h["sample_interval"],
        sound_speed=h["sound_speed_indicative"], absorption=h["absorption_indicative"],
        transmit_power=h["transmit_power"], tau_nominal=h["transmit_duration_nominal"], gain=gain, sa_correction=sa,
        psi=h["equivalent_beam_angle"], f_nominal=h["frequency_nominal"],
        tau_eff=d["transmit_duration_nominal"][:, 0].cpu().numpy())
    return h, sv, er


def _close(got, exp, tol, what):
    got, exp = np.asarray(got, np.float64), np.asarray(exp, np.float64)
    np.testing.assert_array_equal(np.isnan(got), np.isnan(exp), err_msg=what)
    f = np.isfinite(exp)
    if f.any():
        err = np.max(np.abs(got[f] - exp[f]) / np.maximum(np.abs(exp[f]), 1.0))
        assert err < tol, (what, err)


# ---- configs[2]: the chain at 4 x 500 000 x 2000 --------------------------------------------------------------------
@pytest.fixture(scope="module")
def chain():
    torch = _need(200)
    from echopype_amd import ops, synth

    C, P, S = 4, 500_000, 2000
    d = synth.ek60_device(C, P, S)
    coef = _coef(ops, d, d["transmit_duration_nominal"][:, 0].contiguous())
    a2 = coef[..., 4].contiguous()
    n_t = P // 20
    bs = ops.time_bin_offsets(d["ping_time_ns"], int(d["ping_time_ns"][0].item()), BIN_NS, n_t)
    n_r = len(np.arange(0, float((S - 1) * 2.56e-4 * float(d["sound_speed_indicative"].max()) / 2) + 1.0, 1.0)) - 1
    out = {}
    for dt in (torch.float64, torch.float32):
        sv, _, nz = ops.sv_noise_fused(d["backscatter_r"], coef, a2, 20, 50, dtype=dt)
        res = ops.sv_denoise_mvbs(d["backscatter_r"], coef, a2, nz, 20, 3.0, bs, n_t, 1.0, n_r, dtype=dt,
                                  want_noise=dt == torch.float64)
        if dt == torch.float64:
            out["f64"] = dict(Sv=sv, noise=nz, **{k: res[k] for k in ("Sv_noise", "Sv_corrected", "MVBS")})
        else:  # keep only what the fp32-vs-fp64 check needs
            out["f32"] = dict(Sv_corrected=res["Sv_corrected"], MVBS=res["MVBS"])
        del sv, res
    torch.cuda.synchronize()
    yield dict(torch=torch, d=d, n_r=n_r, **out)
    out.clear()
    del d
    gc.collect()
    torch.cuda.empty_cache()


@pytest.mark.parametrize("p0", [0, 251_780, 499_000])
def test_chain_windows_match_the_oracle_chain(chain, p0):
    """1000 pings = 50 whole noise blocks (20 pings) and 50 whole time bins: Sv_noise, Sv_corrected and the MVBS of the
    corrected Sv of the full-size two-pass chain == the oracle's three calls on that window."""
    w = 1000
    h, sv, er = _oracle_window(chain["d"], p0, w)
    exp_n, exp_c = oclean.remove_background_noise(sv, er, h["absorption_indicative"], 20, 50, None, "3.0dB")
    f = chain["f64"]
    _close(f["Sv"][:, p0:p0 + w].cpu().numpy(), sv, 1e-9, "Sv")
    got_n = f["Sv_noise"][:, p0:p0 + w].cpu().numpy()
    # the raw-fed pass keeps Sv_noise finite under NaN-padded samples (documented in the header); compare where defined
    _close(np.where(np.isnan(exp_n), np.nan, got_n), exp_n, 1e-9, "Sv_noise")
    _close(f["Sv_corrected"][:, p0:p0 + w].cpu().numpy(), exp_c, 1e-7, "Sv_corrected")
    pt = chain["d"]["ping_time"][p0:p0 + w]
    exp_m = ogrid.groupby_mean(exp_c, er, pt, ogrid.ping_edges(pt, "20s"), np.arange(0, chain["n_r"] + 1.0, 1.0))
    _close(f["MVBS"][:, p0 // 20:(p0 + w) // 20].cpu().numpy(), exp_m, 1e-7, "MVBS of Sv_corrected")


def test_chain_fp32_vs_fp64_on_the_corrected_output(chain):
    """BASELINE configs[2] 'fp32 vs fp64 tolerance check': where both keep a sample, the corrected Sv agrees to 1e-3
    (relative, dB); the keep / remove decision (SNR threshold, positive difference) may flip only for samples within
    float32 rounding of the threshold -- a vanishing fraction."""
    torch = chain["torch"]
    a, b = chain["f64"]["Sv_corrected"], chain["f32"]["Sv_corrected"]
    flips = total = 0
    worst = 0.0
    for c in range(a.shape[0]):
        na, nb = torch.isnan(a[c]), torch.isnan(b[c])
        flips += int((na != nb).sum().item())
        total += a[c].numel()
        both = ~(na | nb)
        diff = ((a[c] - b[c].double()).abs() / a[c].abs().clamp_min(1.0))
        worst = max(worst, float(torch.where(both, diff, torch.zeros_like(diff)).max()))
    assert worst < 1e-3, worst
    # a flip needs corr - Sv_noise within float32 rounding (~1e-4 dB) of the 3 dB threshold: a few per million
    assert flips / total < 1e-4, (flips, total)
    m64, m32 = chain["f64"]["MVBS"], chain["f32"]["MVBS"].double()
    # a cell whose only surviving sample flipped may appear / vanish: a handful among 39 M cells
    assert int((torch.isnan(m64) != torch.isnan(m32)).sum().item()) < 1e-5 * m64.numel()
    # ... and a cell that keeps only a few samples moves when one of them flips: all but a handful agree to 1e-3
    rel = torch.nan_to_num((m64 - m32).abs() / m64.abs().clamp_min(1.0))
    assert int((rel > 1e-3).sum().item()) < 1e-4 * m64.numel(), float(rel.max())
    assert float(rel.median()) < 1e-5


# ---- configs[3]: EK80 broadband at 2 x 200 000 x 8192 x 4 ------------------------------------------------------------
def _bb_volume(torch, P, plane_dtype, out_dtypes):
    """EK80 BB planes (C, P, 8192, 4) of ``plane_dtype`` generated in HBM + the Sv of the LDS-FFT path."""
    from echopype_amd import _lib, ops, synth
    from oracle import ek80 as oek

    C, S, B = 2, 8192, 4
    g = torch.Generator(device="cuda")
    g.manual_seed(20260504)
    re = torch.empty((C, P, S, B), dtype=plane_dtype, device="cuda")
    im = torch.empty((C, P, S, B), dtype=plane_dtype, device="cuda")
    for p0 in range(0, P, 4000):  # noise + a strong layer that moves with the ping: 60 dB of in-tile dynamic range
        n = min(4000, P - p0)
        for t in (re, im):
            t[:, p0:p0 + n] = torch.randn((C, n, S, B), generator=g, device="cuda", dtype=torch.float32) * 1e-3
    filt, par = synth.ek80_filters(), synth.EK80_BB
    reps = [oek.transmit_replica(1.5e6, par["tau"][c], 0.05, par["f_start"][c], par["f_stop"][c], filt)[0] for c in range(C)]
    # a strong replica-shaped echo (a sea-floor return) that wanders with the ping: ~60 dB above the noise after
    # compression, inside one tile with it
    layer = (torch.arange(P, device="cuda") * 7) % (S - 600) + 100
    ar = torch.arange(P, device="cuda")[:, None]
    for c in range(C):
        m = reps[c].size
        idx = layer[:, None] + torch.arange(m, device="cuda")[None, :]
        rr = torch.from_numpy(np.ascontiguousarray(reps[c].real, dtype=np.float32)).cuda()
        ri = torch.from_numpy(np.ascontiguousarray(reps[c].imag, dtype=np.float32)).cuda()
        amp = 1.0 / float(np.linalg.norm(reps[c]))  # compressed peak = amp, noise floor = 1e-3 / ||tx||: 60 dB
        for b in range(B):
            re[c, ar, idx, b] += (amp * rr[None, :]).to(plane_dtype)
            im[c, ar, idx, b] += (amp * ri[None, :]).to(plane_dtype)
    nan_pings = torch.rand(P, generator=g, device="cuda") < 0.10
    re[:, nan_pings, S - 410:] = float("nan")
    im[:, nan_pings, S - 410:] = float("nan")
    rep = np.concatenate(reps).astype(np.complex64)
    repf = torch.from_numpy(np.ascontiguousarray(rep.view(np.float32))).cuda()
    off = torch.from_numpy(np.cumsum([0] + [r.size for r in reps]).astype(np.int32)).cuda()
    cc = np.zeros((C, P, _lib.NCCOEF))
    cc[..., _lib.CC_RA], cc[..., _lib.CC_RB] = 8e-6, 750.0 + 0.25 * np.sin(np.arange(P) / 1e4)[None, :]
    cc[..., _lib.CC_SHIFT], cc[..., _lib.CC_ALPHA2], cc[..., _lib.CC_A] = 0.19, 0.02, -30.0
    cc[..., _lib.CC_PSCALE] = B / 8.0 * (abs(5400.0 + 75.0) / 5400.0) ** 2 / 75.0
    ccd = torch.from_numpy(cc).cuda()
    out = {}
    for name, dt in out_dtypes:
        out[name] = ops.sv_complex(re, im, ccd, replica=repf, replica_off=off, max_taps=max(r.size for r in reps),
                                   dtype=dt, want_range=False)["out"]
    torch.cuda.synchronize()
    return dict(torch=torch, ops=ops, re=re, im=im, cc=cc, reps=reps, repf=repf, off=off, B=B, S=S, **out)


@pytest.fixture(scope="module")
def bb():
    torch = _need(150)
    d = _bb_volume(torch, 200_000, torch.float32, (("f64", torch.float64), ("f32", torch.float32)))
    yield d
    d.clear()
    gc.collect()
    torch.cuda.empty_cache()


def _bb_oracle_window(bb, p0, w):
    from oracle import ek80 as oek

    x = (bb["re"][:, p0:p0 + w].cpu().numpy().astype(np.float64) + 1j * bb["im"][:, p0:p0 + w].cpu().numpy()).astype(np.complex64)
    prx = oek.power_from_complex(x, 5400.0, 75.0, bb["reps"])  # (C, w, S)
    cc = bb["cc"][:, p0:p0 + w]
    with np.errstate(invalid="ignore", divide="ignore"):
        R = (np.arange(bb["S"])[None, None, :] * cc[..., 0:1]) * cc[..., 1:2]
        rt = R - cc[..., 2:3]
        rt = np.where(rt > 0, rt, np.nan)
        prx = np.where(prx > 0, prx, np.nan)
        exp = 10 * np.log10(prx) + 20 * np.log10(rt) + cc[..., 3:4] * rt + cc[..., 4:5]
    return exp, prx


def test_bb_float64_planes_volume_windows_match_the_oracle():
    """backscatter_r / _i as the converter stores them -- float64 (convert/parse_base.py:306-309) -- at 2 x 50 000 x
    8192 x 4 (52 GB of planes, a quarter of configs[3]'s ping count: the full volume is 210 GB and would not fit next to
    the float32 fixture of this module): windows at the start, in the middle and at the end against the scipy-convolve
    oracle, complex128 transform -> float64 Sv and complex64 -> float32."""
    torch = _need(70)
    d = _bb_volume(torch, 50_000, torch.float64, (("f64", torch.float64), ("f32", torch.float32)))
    try:
        for p0 in (0, 31_313, 49_990):
            exp, prx = _bb_oracle_window(d, p0, 10)
            assert_bb_close(d["f64"][:, p0:p0 + 10].cpu().numpy(), exp, "float64", prx=prx)
            assert_bb_close(d["f32"][:, p0:p0 + 10].cpu().numpy(), exp, "float32", prx=prx)
    finally:
        d.clear()
        gc.collect()
        torch.cuda.empty_cache()


@pytest.mark.parametrize("p0", [0, 77_777, 199_990])
def test_bb_windows_match_the_scipy_convolve_oracle(bb, p0):
    """10 pings of the 2 x 200 000 x 8192 x 4 volume: pulse compression + sector mean + Sv == oracle.ek80 (the
    reference's scipy.signal.convolve loop) + the Sv chain, in linear power against the peak-relative bound of
    tests/bb_tolerance.py (complex128 and complex64 transforms)."""
    w = 10
    exp, prx = _bb_oracle_window(bb, p0, w)
    assert_bb_close(bb["f64"][:, p0:p0 + w].cpu().numpy(), exp, "float64", prx=prx)
    assert_bb_close(bb["f32"][:, p0:p0 + w].cpu().numpy(), exp, "float32", prx=prx)


# ---- configs[4]'s range depth: the fused kernel with 4096 samples and 787 range bins ---------------------------------
@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_fused_kernel_at_the_cfg5_range_depth(dtype):
    torch = _need(20)
    from echopype_amd import ops, synth

    C, P, S = 4, 20_000, 4096
    d = synth.ek60_device(C, P, S, seed=20260505)
    dt = getattr(torch, dtype)
    coef = _coef(ops, d, d["transmit_duration_nominal"][:, 0].contiguous())
    n_t = P // 20
    bs = ops.time_bin_offsets(d["ping_time_ns"], int(d["ping_time_ns"][0].item()), BIN_NS, n_t)
    n_r = len(np.arange(0, float((S - 1) * 2.56e-4 * 1500.5 / 2) + 1.0, 1.0)) - 1
    assert n_r == 787
    res = ops.sv_mvbs_fused(d["backscatter_r"], coef, bs, n_t, 1.0, n_r, dtype=dt, want_range_max=True)
    tol = 1e-9 if dtype == "float64" else 1e-3
    for p0 in (0, 9_980, 19_000):
        w = 1000
        h, sv, er = _oracle_window(d, p0, w)
        _close(res["Sv"][:, p0:p0 + w].cpu().numpy(), sv, tol, "Sv")
        pt = d["ping_time"][p0:p0 + w]
        exp = ogrid.groupby_mean(sv, er, pt, ogrid.ping_edges(pt, "20s"), np.arange(0, n_r + 1.0, 1.0))
        _close(res["MVBS"][:, p0 // 20:(p0 + w) // 20].cpu().numpy(), exp, tol, "MVBS")
    # the range-maximum by-product sizes exactly this grid
    assert len(np.arange(0, float(res["range_max"].item()) + 1.0, 1.0)) - 1 == n_r
