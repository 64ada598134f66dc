"""Parity at BASELINE.json's FULL sizes through size-independent properties (the oracle cannot run
4 G samples): random windows of the full-size result against the oracle, fused == unfused,
count conservation, dB-offset linearity, fp32 vs fp64 tolerance, matched-filter impulse response.
Needs an MI355X with its 288 GB (cfg2 holds ~115 GB of buffers here)."""
import numpy as np
import pytest

from oracle import calibrate as ocal
from oracle import commongrid as ogrid

pytestmark = pytest.mark.gpu

C, P, S = 4, 500_000, 2000  # BASELINE configs[1]
BIN_NS = 20_000_000_000


@pytest.fixture(scope="module")
def vol():
    import torch

    if not torch.cuda.is_available():
        pytest.fail("needs a GPU")
    import gc

    gc.collect()
    torch.cuda.empty_cache()
    free, _ = torch.cuda.mem_get_info()
    if free < 150 * 2**30:
        pytest.skip("needs ~150 GB of free HBM")
    from echopype_amd import ops, synth

    d = synth.ek60_device(C, P, S)
    tau0 = d["transmit_duration_nominal"][:, 0].contiguous()

    def coef(gain=None):
        return ops.power_coef_ek(
            d["sample_interval"], d["transmit_duration_nominal"], d["transmit_power"],
            d["sound_speed_indicative"], d["absorption_indicative"],
            d["gain_correction"] if gain is None else gain, d["sa_correction"], d["equivalent_beam_angle"],
            d["frequency_nominal"], tau0, pulse_length=d["pulse_length"], gain_is_table=True, sa_is_table=True)

    ns = d["ping_time_ns"]
    t0 = int(ns[0].item())
    n_t = P // 20
    bs = ops.time_bin_offsets(ns, t0, BIN_NS, n_t)
    n_r = len(np.arange(0, float((S - 1) * 2.56e-4 * float(d["sound_speed_indicative"].max()) / 2) + 1.0, 1.0)) - 1
    res = ops.sv_mvbs_fused(d["backscatter_r"], coef(), bs, n_t, 1.0, n_r, want_partials=True)
    torch.cuda.synchronize()
    return dict(torch=torch, ops=ops, d=d, coef=coef, bs=bs, n_t=n_t, n_r=n_r, res=res)


def _nan_equal(torch, a, b):
    return all(bool((torch.isnan(a[c]) == torch.isnan(b[c])).all()) for c in range(a.shape[0]))


def _max_err(torch, a, b, shift=0.0, relative=False):
    """max |a - b - shift| (optionally / max(|a|, 1)) over non-NaN a, channel by channel (the boolean
    mask indexing of a 4 G-element tensor is not usable)."""
    worst = 0.0
    for c in range(a.shape[0]):
        diff = (a[c] - b[c].to(a.dtype) - shift).abs()
        if relative:
            diff = diff / a[c].abs().clamp_min(1.0)
        worst = max(worst, float(torch.nan_to_num(diff, nan=0.0).max()))
    return worst


def _host_window(v, p0, w):
    d = v["d"]
    h = {k: d[k][:, p0:p0 + w].cpu().numpy() for k in ("sample_interval", "transmit_duration_nominal", "transmit_power",
                                                       "sound_speed_indicative", "absorption_indicative")}
    for k in ("gain_correction", "sa_correction", "pulse_length", "equivalent_beam_angle", "frequency_nominal"):
        h[k] = d[k].cpu().numpy()
    h["backscatter_r"] = d["backscatter_r"][:, p0:p0 + w].cpu().numpy()
    h["ping_time"] = d["ping_time"][p0:p0 + w]
    h["tau0"] = d["transmit_duration_nominal"][:, 0].cpu().numpy()
    return h


@pytest.mark.parametrize("p0", [0, 123_460, 499_000])
def test_fullsize_windows_match_oracle(vol, p0):
    """Any 1000-ping window (50 whole time bins) of the 4 x 500k x 2000 result == oracle on that window."""
    w = 1000
    h = _host_window(vol, p0, w)
    gain = ocal.vend_cal_params_power(h["transmit_duration_nominal"], h["pulse_length"], h["gain_correction"])
    sa = ocal.vend_cal_params_power(h["transmit_duration_nominal"], h["pulse_length"], h["sa_correction"])
    sv, er = ocal.cal_power_ek(
        h["backscatter_r"], sonar="EK60", cal_type="Sv", sample_interval=h["sample_interval"],
        sound_speed=h["sound_speed_indicative"], absorption=h["absorption_indicative"],
        transmit_power=h["transmit_power"], tau_nominal=h["transmit_duration_nominal"], gain=gain,
        sa_correction=sa, psi=h["equivalent_beam_angle"], f_nominal=h["frequency_nominal"], tau_eff=h["tau0"])
    got = vol["res"]["Sv"][:, p0:p0 + w].cpu().numpy()
    np.testing.assert_array_equal(np.isnan(got), np.isnan(sv))
    f = np.isfinite(sv)
    assert np.max(np.abs(got[f] - sv[f]) / np.maximum(np.abs(sv[f]), 1.0)) < 1e-9
    # MVBS rows of the window: same global range grid
    r_edges = np.arange(0, vol["n_r"] + 1.0, 1.0)
    t_edges = ogrid.ping_edges(h["ping_time"], "20s")
    exp = ogrid.groupby_mean(sv, er, h["ping_time"], t_edges, r_edges)
    b0 = p0 // 20
    gotm = vol["res"]["MVBS"][:, b0:b0 + w // 20].cpu().numpy()
    np.testing.assert_array_equal(np.isnan(gotm), np.isnan(exp))
    f = np.isfinite(exp)
    assert np.max(np.abs(gotm[f] - exp[f]) / np.maximum(np.abs(exp[f]), 1.0)) < 1e-9


def test_fullsize_fused_equals_unfused_and_counts_are_conserved(vol):
    torch, ops, d = vol["torch"], vol["ops"], vol["d"]
    sv1, rng1 = ops.sv_power(d["backscatter_r"], vol["coef"]())
    # same arithmetic in both kernels -> bit-identical Sv (NaN == NaN)
    a, b = vol["res"]["Sv"], sv1
    assert all(bool(((a[c] == b[c]) | (torch.isnan(a[c]) & torch.isnan(b[c]))).all()) for c in range(C))
    mv2 = ops.mvbs(sv1, vol["bs"], vol["n_t"], 1.0, vol["n_r"], range=rng1, want_partials=True)
    m1, m2 = vol["res"]["MVBS"], mv2["MVBS"]
    assert _nan_equal(torch, m1, m2)
    assert _max_err(torch, m1, m2, relative=True) < 1e-11
    assert bool((vol["res"]["cnt"] == mv2["cnt"]).all())
    # every sample with a valid Sv and an in-grid range is counted exactly once
    n_valid = sum(int((~torch.isnan(sv1[c]) & (rng1[c] < float(vol["n_r"]))).sum().item()) for c in range(C))
    assert int(vol["res"]["cnt"].to(torch.int64).sum().item()) == n_valid
    del mv2
    # the route compute_Sv / compute_MVBS take when echo_range stays lazy, at full size: the statistics K1 leaves
    # without writing the array, the array epa_range_power writes on demand, the reduction on the coefficient rows
    cf = vol["coef"]()
    sv2, none, stats = ops.sv_power(d["backscatter_r"], cf, want_range=False, want_range_stats=True)
    assert none is None and all(bool(((sv2[c] == sv1[c]) | (torch.isnan(sv2[c]) & torch.isnan(sv1[c]))).all()) for c in range(C))
    del sv2
    lo, hi, nn = ops.nanminmax(rng1, with_nan_count=True)
    assert stats.cpu().tolist() == [lo, hi, float(nn)]
    rng2 = ops.range_power(d["backscatter_r"], cf)
    assert all(bool(((rng2[c] == rng1[c]) | (torch.isnan(rng2[c]) & torch.isnan(rng1[c]))).all()) for c in range(C))
    del rng2
    mv3 = ops.mvbs(sv1, vol["bs"], vol["n_t"], 1.0, vol["n_r"], coef=cf, coef_as_stored=True, want_partials=True)
    assert _nan_equal(torch, m1, mv3["MVBS"]) and _max_err(torch, m1, mv3["MVBS"], relative=True) < 1e-11
    assert bool((vol["res"]["cnt"] == mv3["cnt"]).all())
    del sv1, rng1, mv3


def test_fullsize_gain_offset_is_a_pure_db_shift(vol):
    """Linearity in the dB domain: +1.5 dB of gain moves every Sv and every MVBS cell by -3 dB."""
    torch, ops, d = vol["torch"], vol["ops"], vol["d"]
    res = ops.sv_mvbs_fused(d["backscatter_r"], vol["coef"](d["gain_correction"] + 1.5), vol["bs"], vol["n_t"], 1.0,
                            vol["n_r"])
    for k in ("Sv", "MVBS"):
        a, b = vol["res"][k], res[k]
        assert _nan_equal(torch, a, b)
        assert _max_err(torch, a, b, shift=3.0) < 1e-9
    del res


def test_fullsize_fp32_vs_fp64_tolerance(vol):
    """BASELINE configs[2]: the fp32 path stays within 1e-3 (relative, dB) of the fp64 path."""
    torch, ops, d = vol["torch"], vol["ops"], vol["d"]
    res = ops.sv_mvbs_fused(d["backscatter_r"], vol["coef"](), vol["bs"], vol["n_t"], 1.0, vol["n_r"],
                            dtype=torch.float32)
    for k in ("Sv", "MVBS"):
        a, b = vol["res"][k], res[k]
        assert _nan_equal(torch, a, b)
        err = _max_err(torch, a, b, relative=True)
        assert err < 1e-3, (k, err)
    del res


@pytest.mark.parametrize("method", ["direct", "fft"])
def test_fullrange_matched_filter_impulse_response(method):
    """EK80 BB at the full range depth of configs[3] (S = 8192): a ping that contains only the
    replica at sample k0 compresses to a unit peak at k0 (|y| / ||tx||^2 == 1), i.e.
    prx(k0) = PSCALE; and the kernel is linear: the sector-sum path equals the per-sector path."""
    import torch

    from echopype_amd import _lib, ops
    from oracle import ek80 as oek

    filt = dict(wbt_fil=(np.hanning(47) * np.exp(2j * np.pi * 0.045 * np.arange(47)) / 10).astype(np.complex64),
                wbt_decifac=6, pc_fil=(np.hanning(91) * np.exp(2j * np.pi * 0.13 * np.arange(91)) / 20).astype(np.complex64),
                pc_decifac=2)
    rep, _ = oek.transmit_replica(1.5e6, 1.024e-3, 0.05, 45e3, 90e3, filt)
    Cc, Pp, Ss, B = 1, 64, 8192, 4
    x = np.zeros((Cc, Pp, Ss, B), dtype=np.complex128)
    rng = np.random.default_rng(0)
    k0s = rng.integers(0, 7000, size=Pp)  # clear of sample 8000, which the mixed-NaN variant blanks
    for p, k0 in enumerate(k0s):
        x[0, p, k0:k0 + rep.size, :] = rep[:, None]
    re, im = np.ascontiguousarray(x.real), np.ascontiguousarray(x.imag)
    cc = np.zeros((Cc, Pp, _lib.NCCOEF))
    cc[..., _lib.CC_RA], cc[..., _lib.CC_RB] = 8e-6, 750.0
    cc[..., _lib.CC_PSCALE] = 1.0
    dev = lambda a: torch.from_numpy(a).cuda()  # noqa: E731
    repf = dev(np.ascontiguousarray(rep.astype(np.complex64).view(np.float32)))
    off = dev(np.array([0, rep.size], dtype=np.int32))
    res = ops.sv_complex(dev(re), dev(im), dev(cc), replica=repf, replica_off=off, max_taps=rep.size,
                         want_prx=True, method=method)
    prx = res["prx"].cpu().numpy()[0]
    peak = prx[np.arange(Pp), k0s]
    np.testing.assert_allclose(peak, 1.0, rtol=1e-6)           # replica stored as complex64
    assert (np.nanargmax(np.nan_to_num(prx, nan=-1.0), axis=1) == k0s).all()
    # mixed-NaN tile takes the per-sector path; removing one sector at one far-away sample must not
    # change the peak (it is zero there anyway)
    re2 = re.copy()
    re2[0, :, 8000, 2] = np.nan
    res2 = ops.sv_complex(dev(re2), dev(im), dev(cc), replica=repf, replica_off=off, max_taps=rep.size,
                          want_prx=True, method=method)
    prx2 = res2["prx"].cpu().numpy()[0]
    np.testing.assert_allclose(prx2[np.arange(Pp), k0s], peak, rtol=1e-12)
    # outside the replica's footprint the correlation is exactly 0 -> prx NaN (calibrate_ek.py:571-575);
    # the FFT form must restore those exact zeros instead of leaving rounding noise of the peak
    k = np.arange(Ss)[None, :]
    outside = (k >= k0s[:, None] + rep.size) | (k + rep.size <= k0s[:, None])
    assert np.isnan(prx[outside]).all() and np.isfinite(prx[~outside]).sum() > Pp * rep.size
