"""Drop-in API parity on a real MI355X: echopype_amd.{calibrate,clean,commongrid} called exactly
like the reference's functions (Dataset / EchoData in, Dataset out), compared with the CPU oracle.

These read like the reference's own tests (tests/calibrate/test_calibrate.py, tests/commongrid/
test_commongrid_api.py, tests/clean/test_noise.py) with the downloaded instrument files replaced
by the seeded synthetic generators of echopype_amd.synth.
"""
import logging

import numpy as np
import pytest

import kat_fixtures as kf
import oracle_chain as oc
from oracle import clean as oclean
from oracle import commongrid as ogrid

from bb_tolerance import assert_bb_close  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ep():
    import torch

    if not torch.cuda.is_available():
        pytest.fail("these tests need a GPU")
    import echopype_amd

    return echopype_amd


def close(got, exp, rtol, what=""):
    got, exp = np.asarray(got, dtype=np.float64), np.asarray(exp, dtype=np.float64)
    assert got.shape == exp.shape, (what, got.shape, exp.shape)
    np.testing.assert_array_equal(np.isnan(got), np.isnan(exp), err_msg=f"{what}: NaN pattern")
    fin = np.isfinite(exp)
    err = np.abs(got[fin] - exp[fin]) / np.maximum(np.abs(exp[fin]), 1.0)
    assert err.size == 0 or err.max() <= rtol, f"{what}: max rel err {err.max():.3e} > {rtol}"


def sv_dataset(ep, d, extra=None):
    """An Sv dataset as a user would hold it (host arrays)."""
    C, P, S = d["Sv"].shape
    ds = ep.Dataset(coords={"channel": [f"ch_{i}" for i in range(C)], "ping_time": d["ping_time"],
                            "range_sample": np.arange(S)})
    dims = ("channel", "ping_time", "range_sample")
    ds["Sv"] = (dims, d["Sv"])
    ds["echo_range"] = (dims, d["echo_range"])
    if "depth" in d:
        ds["depth"] = (dims, d["depth"])
    ds["frequency_nominal"] = (("channel",), np.arange(C, dtype=float))
    for k, v in (extra or {}).items():
        ds[k] = v
    return ds


# ------------------------------------------------------------------------------------ calibrate
@pytest.mark.parametrize("dtype,rtol", [("float64", 1e-9), ("float32", 1e-3)])
@pytest.mark.parametrize("cal", ["Sv", "TS"])
def test_compute_Sv_TS_ek60(ep, dtype, rtol, cal):
    d = ep.synth.ek60_numpy(2, 120, 1000, vary_tau=True)
    ed = ep.echodata.from_ek60_arrays(d)
    fn = ep.calibrate.compute_Sv if cal == "Sv" else ep.calibrate.compute_TS
    ds = fn(ed, dtype=dtype)
    exp, exp_r = oc.ek60(d, cal)
    close(ds[cal].values, exp, rtol, cal)
    if dtype == "float64":
        np.testing.assert_array_equal(ds["echo_range"].values, exp_r)
        assert ds[cal].dtype == np.float64
    assert ds[cal].dims == ("channel", "ping_time", "range_sample")
    assert ds[cal].attrs["units"] == "dB"
    assert ds.attrs["processing_function"] == f"calibrate.compute_{cal}"
    for k in ("sound_speed", "sound_absorption", "gain_correction", "sa_correction", "equivalent_beam_angle",
              "frequency_nominal"):
        assert k in ds, k
    if cal == "Sv":  # EK60: tau_effective == transmit_duration_nominal of ping 0 (calibrate_ek.py:134-151)
        np.testing.assert_array_equal(ds["tau_effective"].values, d["transmit_duration_nominal"][:, 0])


def test_compute_Sv_ek60_user_env_and_cal_params(ep):
    d = ep.synth.ek60_numpy(2, 40, 512)
    ed = ep.echodata.from_ek60_arrays(d)
    env = {"temperature": 8.0, "salinity": 34.0, "pressure": 50.0, "pH": 8.05}
    gain = [25.1, 26.3]
    ds = ep.calibrate.compute_Sv(ed, env_params=env, cal_params={"gain_correction": gain})
    exp, _ = oc.ek60(d, "Sv", env=env, gain=np.tile(np.array(gain)[:, None], (1, 40)))
    close(ds["Sv"].values, exp, 1e-9, "user params")
    assert str(ds["formula_absorption"].values) == "FG"


def test_compute_Sv_argument_errors(ep):
    d = ep.synth.ek60_numpy(1, 4, 64)
    ed = ep.echodata.from_ek60_arrays(d)
    with pytest.raises(ValueError, match="assume_single_filter_time can only be used on complex EK80 data."):
        ep.calibrate.compute_Sv(ed, assume_single_filter_time=True)
    ed80 = ep.echodata.from_ek80_arrays(ep.synth.ek80_numpy(1, 2, 64), ep.synth.ek80_filters())
    with pytest.raises(ValueError, match="waveform_mode and encode_mode must be specified for EK80 calibration"):
        ep.calibrate.compute_Sv(ed80)
    with pytest.raises(ValueError, match="must be recorded as complex samples"):
        ep.calibrate.compute_Sv(ed80, waveform_mode="BB", encode_mode="power")
    with pytest.raises(RuntimeError, match="No beam group with the specified encode_mode"):
        ep.calibrate.compute_Sv(ed80, waveform_mode="CW", encode_mode="power")
    ed80["Sonar"] = ep.Dataset()
    with pytest.raises(ValueError, match="Echodata missing `waveform_encode_descr`"):
        ep.calibrate.compute_Sv(ed80, waveform_mode="BB", encode_mode="complex")
    with pytest.raises(ValueError, match="sound_absorption"):
        ep.calibrate.compute_Sv(ed, env_params={"sound_absorption": 0.01})


@pytest.mark.parametrize("cal", ["Sv", "TS"])
def test_compute_Sv_TS_azfp(ep, cal):
    d = ep.synth.azfp_numpy(4, 30, 500)
    ed = ep.echodata.from_azfp_arrays(d)
    fn = ep.calibrate.compute_Sv if cal == "Sv" else ep.calibrate.compute_TS
    ds = fn(ed, env_params={"salinity": d["salinity"], "pressure": d["pressure"]})
    exp, exp_r = oc.azfp(d, cal)
    close(ds[cal].values, exp, 1e-9, f"AZFP {cal}")
    close(ds["echo_range"].values, exp_r, 1e-12, "AZFP range")
    with pytest.raises(ReferenceError, match="Please supply both salinity and pressure"):
        fn(ed)


def _ek80(ep, waveform, **kw):
    filt = ep.synth.ek80_filters()
    wf = "BB" if waveform == "BB" else "CW"
    d0 = dict(ep.synth.EK80_BB)
    from oracle_chain import ek80_replicas
    probe = {**{k: v[:2] for k, v in d0.items()}, "fs": np.full(2, 1.5e6), "slope": np.full(2, 0.05)}
    reps, _ = ek80_replicas(probe, filt, wf)
    d = ep.synth.ek80_numpy(waveform=wf, replicas=reps if wf == "BB" else None, **kw)
    return d, filt


@pytest.mark.parametrize("dtype", ["float64", "float32"])
@pytest.mark.parametrize("mixed", [False, True])
def test_compute_Sv_ek80_bb(ep, dtype, mixed):
    """BB pulse compression + Sv.  The reference's own output is complex64-rounded
    (ek80_complex.py:304), so parity is judged like the reference's tests do: in dB with an
    absolute tolerance (test_calibrate_ek80_CW.py uses 2e-3...5.5e-3 dB), see tests/bb_tolerance.py, plus
    identical NaN patterns."""
    d, filt = _ek80(ep, "BB", C=2, P=12, S=1200, mixed_nan=mixed)
    ed = ep.echodata.from_ek80_arrays(d, filt)
    ds = ep.calibrate.compute_Sv(ed, waveform_mode="BB", encode_mode="complex", dtype=dtype)
    (exp, exp_r, prx), teff = oc.ek80_complex(d, filt, "Sv")
    assert_bb_close(ds["Sv"].values, exp, dtype, prx=prx)
    np.testing.assert_allclose(ds["tau_effective"].values, teff, rtol=1e-12)
    if dtype == "float64":
        np.testing.assert_array_equal(ds["echo_range"].values, exp_r)
    assert ds["Sv"].attrs["waveform_mode"] == "BB"


def test_compute_Sv_ek80_multi_filter_time_equals_single(ep):
    """tests/calibrate/test_calibrate_ek80.py:610-666: a file with several filter_time entries gives
    the same Sv whether each (channel, filter interval) is calibrated separately and merged, or a
    single filter set is assumed, or the file holds one filter_time."""
    d, filt = _ek80(ep, "BB", C=2, P=12, S=600)
    single = ep.calibrate.compute_Sv(ep.echodata.from_ek80_arrays(d, filt), waveform_mode="BB", encode_mode="complex")
    ed_multi = ep.echodata.from_ek80_arrays(d, filt, filter_time_idx=[0, 5, 9])
    assert ed_multi["Vendor_specific"].sizes["filter_time"] == 3
    merged = ep.calibrate.compute_Sv(ed_multi, waveform_mode="BB", encode_mode="complex")
    assumed = ep.calibrate.compute_Sv(ed_multi, waveform_mode="BB", encode_mode="complex",
                                      assume_single_filter_time=True)
    for other in (merged, assumed):
        np.testing.assert_array_equal(other["Sv"].values, single["Sv"].values)
        np.testing.assert_array_equal(other["echo_range"].values, single["echo_range"].values)
    assert merged["Sv"].dims == ("channel", "ping_time", "range_sample")
    np.testing.assert_allclose(merged["sound_absorption"].values, single["sound_absorption"].values)


@pytest.mark.parametrize("wf", ["BB", "CW"])
@pytest.mark.parametrize("method", ["auto", "direct"])
def test_compute_Sv_ek80_filter_intervals_with_different_replicas_in_one_launch(ep, wf, method, monkeypatch):
    """A file whose filter_time intervals carry DIFFERENT filters (so replicas of different lengths and effective pulse
    lengths) and whose first pings precede the first filter_time: the reference calibrates every (channel, interval)
    slice on its own and merges with an outer join (calibrate/api.py:125-197).  Here ONE launch covers the grid (a
    replica index per ping, epa_sv_complex[_fft]_indexed) -- held to the slices calibrated one by one as single-filter
    files and merged in NumPy: NaN for the pings no interval covers, Sv / echo_range bit for bit elsewhere."""
    from echopype_amd import _lib, ops

    C, P, S = 2, 14, 2300
    d, filt = _ek80(ep, wf, C=C, P=P, S=S, mixed_nan=True)
    # (an EK80 file records ONE sound speed, from_ek80_arrays takes the first ping's: the same for the file and its slices)
    d["sound_speed"] = np.full_like(d["sound_speed"], d["sound_speed"].flat[0])
    starts = [2, 6, 10]                                   # pings 0, 1 lie before the first filter_time
    k61 = np.arange(61)
    filt2 = dict(filt, pc_fil=(np.hanning(61) * np.exp(2j * np.pi * 0.11 * k61) / 15).astype(np.complex64))
    filts = [filt, filt2, filt]
    ed = ep.echodata.from_ek80_arrays(d, filt, filter_time_idx=starts)
    vend = ed["Vendor_specific"]
    n_pc = vend["PC_coeffs_real"].shape[2]
    for name, part in (("PC_coeffs_real", np.real), ("PC_coeffs_imag", np.imag)):   # the second interval's PC filter
        a = np.array(vend[name].values)
        a[:, 1, :] = np.nan
        a[:, 1, :61] = part(filt2["pc_fil"])
        vend[name] = (vend[name].dims, a)
    assert n_pc >= 61
    if method == "direct":
        monkeypatch.setattr(ops, "sv_complex_uses_fft", lambda *a, **k: False)
    kw = dict(waveform_mode=wf, encode_mode="complex")
    with _lib.launch_trace() as tr:
        got = ep.calibrate.compute_Sv(ed, **kw)
    sample_kernels = [k for k in tr.kernels if k in ("sv_complex_kernel", "sv_complex_fft_kernel", "sv_complex_cw_kernel")]
    assert len(sample_kernels) == 1, tr.kernels          # one launch for all six (channel, interval) pairs
    if wf == "BB":
        assert sample_kernels == ["sv_complex_fft_kernel" if method == "auto" else "sv_complex_kernel"]
    exp_sv = np.full((C, P, S), np.nan)
    exp_r = np.full((C, P, S), np.nan)
    exp_te = np.full((C, P), np.nan)
    bounds = starts + [P]
    per_ping = ("backscatter_r", "backscatter_i", "sample_interval", "sound_speed")
    for ci in range(C):
        for k in range(3):
            a, b = bounds[k], bounds[k + 1]
            dk = {}
            for key, v in d.items():
                if key in per_ping:
                    dk[key] = np.ascontiguousarray(v[ci:ci + 1, a:b])
                elif key == "ping_time":
                    dk[key] = v[a:b]
                elif key == "channel":
                    dk[key] = v[ci:ci + 1]
                elif isinstance(v, np.ndarray) and v.shape[:1] == (C,):
                    dk[key] = v[ci:ci + 1]
                else:
                    dk[key] = v
            one = ep.calibrate.compute_Sv(ep.echodata.from_ek80_arrays(dk, filts[k]), **kw)
            exp_sv[ci, a:b], exp_r[ci, a:b] = one["Sv"].values[0], one["echo_range"].values[0]
            exp_te[ci, a:b] = one["tau_effective"].values[0]
    if wf == "BB" and method == "auto":
        # the overlap-save tiles are laid out for the LONGEST replica of the launch (2048 - max_taps + 1 outputs each):
        # a slice calibrated on its own is tiled for its own replica, so the transforms round differently (~1e-7 dB;
        # the reference's own tolerance for broadband Sv is 2e-3 dB)
        np.testing.assert_array_equal(np.isnan(got["Sv"].values), np.isnan(exp_sv))
        np.testing.assert_allclose(got["Sv"].values, exp_sv, rtol=0, atol=1e-5, equal_nan=True)
    else:
        np.testing.assert_array_equal(got["Sv"].values, exp_sv)
    np.testing.assert_array_equal(got["echo_range"].values, exp_r)
    assert np.isnan(got["Sv"].values[:, :2]).all() and np.isfinite(got["Sv"].values[:, 2:]).any()
    # the second interval's filter gives another effective pulse length: tau_effective goes out per (channel, ping)
    assert got["tau_effective"].dims == ("channel", "ping_time")
    np.testing.assert_array_equal(got["tau_effective"].values, exp_te)
    assert exp_te[0, 3] != exp_te[0, 7] and exp_te[0, 3] == exp_te[0, 11]
    # (channel, ping_time) parameters are NaN where no slice exists (the outer join's fill)
    grid_vars = [n for n, v in got.data_vars.items() if tuple(v.dims) == ("channel", "ping_time") and v.dtype.kind == "f"]
    assert "tau_effective" in grid_vars and (wf == "CW" or "sound_absorption" in grid_vars)
    for n in grid_vars:
        a = got[n].values
        assert np.isnan(a[:, :2]).all() and np.isfinite(a[:, 2:]).all(), n


@pytest.mark.parametrize("cal", ["Sv", "TS"])
def test_compute_Sv_TS_ek80_cw_complex(ep, cal):
    d, filt = _ek80(ep, "CW", C=2, P=10, S=700, mixed_nan=True)
    ed = ep.echodata.from_ek80_arrays(d, filt)
    fn = ep.calibrate.compute_Sv if cal == "Sv" else ep.calibrate.compute_TS
    ds = fn(ed, waveform_mode="CW", encode_mode="complex")
    (exp, exp_r, _), _ = oc.ek80_complex(d, filt, cal)
    close(ds[cal].values, exp, 1e-9, f"EK80 CW complex {cal}")


# ------------------------------------------------------------------------------------ the chain
@pytest.mark.parametrize("dtype,rtol", [("float64", 1e-9), ("float32", 1e-3)])
def test_chain_sv_noise_mvbs(ep, dtype, rtol):
    """compute_Sv -> remove_background_noise -> compute_MVBS, device-resident between the calls."""
    d = ep.synth.ek60_numpy(2, 205, 1000)
    ds = ep.calibrate.compute_Sv(ep.echodata.from_ek60_arrays(d), dtype=dtype)
    assert ep.xr_lite.is_device(ds["Sv"].data)
    out = ep.clean.remove_background_noise(ds, ping_num=20, range_sample_num=50, SNR_threshold="3.0dB")
    assert out is ds and "Sv_noise" in ds and "Sv_corrected" in ds  # in-place (clean/api.py:490-502)
    mv = ep.commongrid.compute_MVBS(ds, range_bin="1m", ping_time_bin="20s")
    sv, er = oc.ek60(d, "Sv")
    exp_n, exp_c = oclean.remove_background_noise(sv, er, d["absorption_indicative"], 20, 50)
    # float32 mode stores echo_range in float32 and compute_MVBS bins the STORED coordinate (as the
    # reference would for a float32 echo_range): feed the oracle the same rounded coordinate, else
    # samples within 1e-7 of a bin edge land in the neighbouring bin -- a discrete effect of the
    # storage precision, not of the arithmetic.
    er_binned = er if dtype == "float64" else er.astype(np.float32).astype(np.float64)
    exp_mv, t_left, r_left = ogrid.compute_MVBS(sv, er_binned, d["ping_time"], "1m", "20s")
    close(ds["Sv_noise"].values, exp_n, rtol, "Sv_noise")
    if dtype == "float64":
        close(ds["Sv_corrected"].values, exp_c, 1e-7, "Sv_corrected")
        assert ds["Sv_corrected"].attrs["actual_range"] == [round(float(np.nanmin(exp_c)), 2),
                                                            round(float(np.nanmax(exp_c)), 2)]
    close(mv["Sv"].values, exp_mv, rtol, "MVBS")
    np.testing.assert_array_equal(mv["ping_time"].values, t_left)
    np.testing.assert_array_equal(mv["echo_range"].values, r_left)
    assert mv["Sv"].attrs["binning_mode"] == "physical units"
    assert mv.attrs["processing_function"] == "commongrid.compute_MVBS"


@pytest.mark.parametrize("edge_case", [False, True])
def test_fused_compute_Sv_MVBS_equals_two_calls(ep, edge_case):
    """The one-pass entry point == compute_Sv followed by compute_MVBS: same Sv, same grid, same
    MVBS; the range grid comes from the kernel's nanmax(echo_range) by-product -- including the
    case where that maximum sits exactly on a bin edge (np.arange drops the last sample)."""
    d = ep.synth.ek60_numpy(2, 130, 800)
    if edge_case:
        d["backscatter_r"] = np.where(np.isnan(d["backscatter_r"]), np.float32(-60), d["backscatter_r"])
        d["sample_interval"][:] = 1.0 / 3000.0            # k = 0.25 m with c = 1500 exactly
        d["sound_speed_indicative"][:] = 1500.0
        rb = "0.25m"
    else:
        rb = "1m"
    ed = ep.echodata.from_ek60_arrays(d)
    ds1 = ep.calibrate.compute_Sv(ed)
    mv1 = ep.commongrid.compute_MVBS(ds1, range_bin=rb, ping_time_bin="20s")
    ds2, mv2 = ep.compute_Sv_MVBS(ed, range_bin=rb, ping_time_bin="20s")
    np.testing.assert_array_equal(ds2["Sv"].values, ds1["Sv"].values)
    assert mv2["Sv"].shape == mv1["Sv"].shape
    np.testing.assert_array_equal(mv2["echo_range"].values, mv1["echo_range"].values)
    np.testing.assert_array_equal(mv2["ping_time"].values, mv1["ping_time"].values)
    close(mv2["Sv"].values, mv1["Sv"].values, 1e-11, "fused vs two calls")
    assert mv2["Sv"].attrs == mv1["Sv"].attrs
    # echo_range of the fused call: lazy (nothing written), the reference's values and attributes when read
    from echopype_amd.xr_lite import LazyDeviceArray
    assert isinstance(ds2["echo_range"].data, LazyDeviceArray) and not ds2["echo_range"].data.materialized
    np.testing.assert_array_equal(ds2["echo_range"].values, ds1["echo_range"].values)
    assert ds2["echo_range"].attrs == ds1["echo_range"].attrs
    # flag combinations the fused kernel does not serve take the two-call route: same answer
    ds3, mv3 = ep.compute_Sv_MVBS(ed, range_bin=rb, ping_time_bin="20s", closed="right")
    mv4 = ep.commongrid.compute_MVBS(ds1, range_bin=rb, ping_time_bin="20s", closed="right")
    close(mv3["Sv"].values, mv4["Sv"].values, 1e-11, "closed=right")
    assert "echo_range" in ds3


@pytest.mark.parametrize("dtype", ["float64", "float32"])
@pytest.mark.parametrize("closed", ["left", "right"])
def test_two_pass_chain_equals_three_calls(ep, dtype, closed):
    """compute_Sv_clean_MVBS (two sweeps of the raw power) == compute_Sv -> remove_background_noise ->
    compute_MVBS of Sv_corrected (four array sweeps), and both == the oracle chain."""
    d = ep.synth.ek60_numpy(2, 205, 1000)
    ed = ep.echodata.from_ek60_arrays(d)
    kw = dict(background_noise_max="-125.0dB", SNR_threshold="3.0dB")
    ds1 = ep.calibrate.compute_Sv(ed, dtype=dtype)
    ep.clean.remove_background_noise(ds1, 20, 50, **kw)
    corr = ds1.copy()
    corr["Sv"] = ds1["Sv_corrected"]
    mv1 = ep.commongrid.compute_MVBS(corr, range_bin="1m", ping_time_bin="20s", closed=closed)
    ds2, mv2 = ep.compute_Sv_clean_MVBS(ed, 20, 50, range_bin="1m", ping_time_bin="20s", closed=closed, dtype=dtype,
                                        materialize_echo_range=True, **kw)
    np.testing.assert_array_equal(ds2["Sv"].values, ds1["Sv"].values)
    np.testing.assert_array_equal(ds2["echo_range"].values, ds1["echo_range"].values)
    rtol = 1e-9 if dtype == "float64" else 1e-3
    close(ds2["Sv_noise"].values, ds1["Sv_noise"].values, 1e-11 if dtype == "float64" else 2e-4, "Sv_noise")
    close(ds2["Sv_corrected"].values, ds1["Sv_corrected"].values, rtol, "Sv_corrected")
    np.testing.assert_array_equal(mv2["echo_range"].values, mv1["echo_range"].values)
    np.testing.assert_array_equal(mv2["ping_time"].values, mv1["ping_time"].values)
    if dtype == "float64":  # (fp32: the three calls bin the float32-stored echo_range, see test_chain_sv_noise_mvbs)
        close(mv2["Sv"].values, mv1["Sv"].values, 1e-9, "MVBS of Sv_corrected")
        for k in ("noise_ping_num", "noise_range_sample_num", "SNR_threshold", "noise_max", "units"):
            assert ds2["Sv_corrected"].attrs[k] == ds1["Sv_corrected"].attrs[k]
        assert ds2["Sv_corrected"].attrs["actual_range"] == ds1["Sv_corrected"].attrs["actual_range"]
        assert mv2["Sv"].attrs["cell_methods"] == mv1["Sv"].attrs["cell_methods"]
        if closed == "left":
            sv, er = oc.ek60(d, "Sv")
            exp_n, exp_c = oclean.remove_background_noise(sv, er, d["absorption_indicative"], 20, 50, "-125.0dB")
            exp_mv, _, _ = ogrid.compute_MVBS(exp_c, er, d["ping_time"], "1m", "20s")
            close(ds2["Sv_noise"].values, exp_n, 1e-9, "oracle Sv_noise")
            close(ds2["Sv_corrected"].values, exp_c, 1e-7, "oracle Sv_corrected")
            close(mv2["Sv"].values, exp_mv, 1e-9, "oracle MVBS")
    # without the materialised echo_range / Sv_noise: same numbers, fewer bytes
    ds3, mv3 = ep.compute_Sv_clean_MVBS(ed, 20, 50, range_bin="1m", ping_time_bin="20s", closed=closed, dtype=dtype,
                                        keep_Sv_noise=False, **kw)
    assert "Sv_noise" not in ds3 and not ds3["echo_range"].data.materialized      # (lazy: nothing written)
    np.testing.assert_array_equal(ds3["echo_range"].values, ds2["echo_range"].values)
    np.testing.assert_array_equal(ds3["Sv_corrected"].values, ds2["Sv_corrected"].values)
    close(mv3["Sv"].values, mv2["Sv"].values, 1e-12 if dtype == "float64" else 1e-5, "rerun")  # LDS atomics order
    with pytest.raises(TypeError, match="Decibal input must be a string"):
        ep.compute_Sv_clean_MVBS(ed, 20, 50, SNR_threshold=3.0)


def test_two_pass_chain_other_sonars_take_the_separate_calls(ep):
    d = ep.synth.azfp_numpy(2, 60, 400)
    ed = ep.echodata.from_azfp_arrays(d)
    env = {"temperature": 8.0, "salinity": 30.0, "pressure": 60.0}
    ds, mv = ep.compute_Sv_clean_MVBS(ed, 10, 40, range_bin="2m", ping_time_bin="10s", env_params=env)
    ref = ep.calibrate.compute_Sv(ed, env_params=env)
    ep.clean.remove_background_noise(ref, 10, 40)
    np.testing.assert_array_equal(ds["Sv_corrected"].values, ref["Sv_corrected"].values)
    assert mv["Sv"].dims == ("channel", "ping_time", "echo_range")


# ------------------------------------------------------------------------------------ commongrid
@pytest.mark.parametrize("kind", ["regular", "irregular"])
def test_compute_MVBS_reference_values(ep, kind, caplog):
    d = kf.mock_small(kind)
    with caplog.at_level(logging.WARNING):
        ds = ep.commongrid.compute_MVBS(sv_dataset(ep, d), range_bin="2m", ping_time_bin="1s")
    exp = kf.brute_force_mvbs(d, "1s", 2)
    assert ds["Sv"].shape == exp.shape
    np.testing.assert_allclose(ds["Sv"].values, exp, atol=1e-10, rtol=1e-10, equal_nan=True)
    if kind == "irregular":  # test_commongrid_api.py:511-519
        assert any("The ```echo_range``` coordinate array contain NaNs." in r.message for r in caplog.records)


def test_compute_MVBS_shapes_edges_positions(ep):
    d = kf.sv_regular()
    P = d["Sv"].shape[1]
    lat = np.linspace(42.48916859, 42.49071833, P)
    lon = np.linspace(-124.88296688, -124.81919229, P)
    ds_in = sv_dataset(ep, d, {"latitude": (("ping_time",), lat), "longitude": (("ping_time",), lon)})
    ds_in.attrs["processing_level"] = "Level 2A"
    ds = ep.commongrid.compute_MVBS(ds_in, range_bin="5m", ping_time_bin="10s")
    dt = (d["ping_time"][-1] - d["ping_time"][0]).astype(np.int64)
    assert ds["Sv"].shape == (2, int(np.ceil(dt / 1e9 / 10)), int(np.ceil(d["echo_range"].max() / 5)))
    exp_mv, t_left, _ = ogrid.compute_MVBS(d["Sv"], d["echo_range"], d["ping_time"], "5m", "10s")
    np.testing.assert_array_equal(ds["ping_time"].values, t_left)
    it = ogrid.bin_index(d["ping_time"], ogrid.ping_edges(d["ping_time"], "10s"))
    np.testing.assert_allclose(ds["latitude"].values, [lat[it == i].mean() for i in range(len(t_left))], rtol=1e-14)
    assert ds.attrs["processing_level"] == "Level 3A"
    mx = ep.commongrid.compute_MVBS(sv_dataset(ep, kf.mock_small("regular")), range_bin="1m", range_var_max="8m")
    assert mx["echo_range"].values.max() == 8  # test_commongrid_api.py:580-592


@pytest.mark.parametrize("skipna,range_var", [(True, "depth"), (False, "depth"), (True, "echo_range"), (False, "echo_range")])
def test_compute_MVBS_skipna_masks(ep, skipna, range_var):
    d = kf.mock_small("irregular")
    sub = {k: (v[:, :2].copy() if v.ndim == 3 else v[:2]) for k, v in d.items()}
    da = ep.commongrid.compute_MVBS(sv_dataset(ep, sub), range_var=range_var, range_bin="2m", skipna=skipna)["Sv"]
    mask = np.isnan(da.values)
    if range_var == "echo_range":
        exp = [[[False] * 5], [[False] * 5]]
    elif skipna:
        exp = [[[True, False, False, False, False, False]]] * 2
    else:
        exp = [[[True, True, True, False, False, True]], [[True, False, False, True, True, True]]]
    np.testing.assert_array_equal(mask, np.array(exp))


def test_compute_MVBS_unsorted_pings_and_closed_right(ep):
    d = kf.sv_regular(2, 60, 0.5, 90, "0.7s")
    perm = np.random.default_rng(1).permutation(90)
    shuf = dict(Sv=d["Sv"][:, perm], echo_range=d["echo_range"][:, perm], ping_time=d["ping_time"][perm])
    for closed in ("left", "right"):
        exp, _, _ = ogrid.compute_MVBS(d["Sv"], d["echo_range"], d["ping_time"], "3m", "5s", closed=closed)
        got = ep.commongrid.compute_MVBS(sv_dataset(ep, shuf), range_bin="3m", ping_time_bin="5s", closed=closed)
        close(got["Sv"].values, exp, 1e-10, f"unsorted closed={closed}")


def test_compute_MVBS_argument_errors(ep):
    ds = sv_dataset(ep, kf.mock_small("regular"))
    with pytest.raises(TypeError, match="range_bin must be a string"):
        ep.commongrid.compute_MVBS(ds, range_bin=10)
    with pytest.raises(ValueError, match="Range bin must be in meters"):
        ep.commongrid.compute_MVBS(ds, range_bin="10km")
    with pytest.raises(TypeError, match="ping_time_bin must be a string"):
        ep.commongrid.compute_MVBS(ds, ping_time_bin=20)
    with pytest.raises(ValueError, match="range_var must be one of 'echo_range' or 'depth'."):
        ep.commongrid.compute_MVBS(ds, range_var="range")
    with pytest.raises(ValueError, match="is not a valid option. Options are 'left' or 'right'."):
        ep.commongrid.compute_MVBS(ds, closed="both")
    for method in ("blockwise", "cohorts"):
        for reindex in (True, False):
            with pytest.raises(ValueError, match=f"Passing in reindex={reindex} is only allowed when method='map_reduce'."):
                ep.commongrid.compute_MVBS(ds, method=method, reindex=reindex)


def test_compute_MVBS_index_binning(ep):
    d = kf.sv_regular(4, 4000, 0.5, 100)
    ds = ep.commongrid.compute_MVBS_index_binning(sv_dataset(ep, d), range_sample_num=7, ping_num=3)
    exp, exp_r = ogrid.compute_MVBS_index_binning(d["Sv"], d["echo_range"], 7, 3)
    assert ds["Sv"].shape == tuple(np.ceil((4, 100 / 3, 4000 / 7)).astype(int))
    close(ds["Sv"].values, exp, 1e-12, "index binning")
    np.testing.assert_array_equal(ds["echo_range"].values, exp_r)
    assert ds["Sv"].attrs["actual_range"] == [round(float(np.nanmin(exp)), 2), round(float(np.nanmax(exp)), 2)]
    np.testing.assert_array_equal(ds["range_sample"].values, np.arange(exp.shape[2]))


@pytest.mark.parametrize("tag", ["ix0", "ix1", "ix2", "ix3"])
def test_compute_MVBS_index_binning_vs_reference_goldens(ep, tag):
    """Outputs of the reference's own compute_MVBS_index_binning (oracle/gen_mvbs_index_goldens.py):
    values, block minimum of echo_range, coarsened ping_time labels, range_sample reset."""
    import os

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_mvbs_index_goldens.npz"))
    rn, pn = (int(v) for v in g[f"{tag}_args"])
    d = {"Sv": g[f"{tag}_Sv"], "echo_range": g[f"{tag}_echo_range"], "ping_time": g[f"{tag}_ping_time"]}
    ds = ep.commongrid.compute_MVBS_index_binning(sv_dataset(ep, d), range_sample_num=rn, ping_num=pn)
    close(ds["Sv"].values, g[f"{tag}_out_Sv"], 1e-12, "index binning vs reference")
    np.testing.assert_array_equal(np.isnan(ds["Sv"].values), np.isnan(g[f"{tag}_out_Sv"]))
    np.testing.assert_array_equal(ds["echo_range"].values, g[f"{tag}_out_echo_range"])
    np.testing.assert_array_equal(ds["ping_time"].values, g[f"{tag}_out_ping_time"])
    np.testing.assert_array_equal(ds["range_sample"].values, g[f"{tag}_out_range_sample"])


def test_echo_range_statistics_travel_with_the_array_until_it_is_modified(ep):
    """compute_Sv leaves nanmin / nanmax / NaN count of echo_range with the device array (a by-product of the
    kernel); compute_MVBS uses them instead of sweeping the array, and stops trusting them once the tensor has
    been modified in place."""
    d = ep.synth.ek60_numpy(2, 60, 500)
    ds = ep.calibrate.compute_Sv(ep.echodata.from_ek60_arrays(d))
    er = ds["echo_range"].values
    st = ds["echo_range"].data.cached_stats()
    assert st == (float(np.nanmin(er)), float(np.nanmax(er)), int(np.isnan(er).sum()))
    mv = ep.commongrid.compute_MVBS(ds, range_bin="5m", ping_time_bin="20s")
    assert mv["echo_range"].values[-1] <= np.nanmax(er) < mv["echo_range"].values[-1] + 5
    ds["echo_range"].data.tensor.mul_(2.0)   # twice the range: the cached maximum is stale now
    assert ds["echo_range"].data.cached_stats() is None
    mv2 = ep.commongrid.compute_MVBS(ds, range_bin="5m", ping_time_bin="20s")
    assert mv2["echo_range"].values[-1] <= 2 * np.nanmax(er) < mv2["echo_range"].values[-1] + 5
    exp, _, _ = ogrid.compute_MVBS(ds["Sv"].values, 2 * er, d["ping_time"], "5m", "20s")
    close(mv2["Sv"].values, exp, 1e-9, "MVBS after the range was doubled in place")


def test_add_depth_then_MVBS_on_depth(ep):
    """compute_Sv -> add_depth(depth_offset, tilt) -> compute_MVBS(range_var="depth") (the reference's
    MVBS value fixtures are built through add_depth, tests/commongrid/conftest.py:101-118)."""
    d = ep.synth.ek60_numpy(2, 61, 600)
    ds = ep.calibrate.compute_Sv(ep.echodata.from_ek60_arrays(d))
    out = ep.consolidate.add_depth(ds, depth_offset=2.5, tilt=15.0)
    assert out is ds
    sv, er = oc.ek60(d, "Sv")
    exp_depth = 2.5 + er * np.cos(np.deg2rad(15.0))
    close(ds["depth"].values, exp_depth, 1e-14, "depth")
    mv = ep.commongrid.compute_MVBS(ds, range_var="depth", range_bin="2m", ping_time_bin="20s")
    exp, t_left, r_left = ogrid.compute_MVBS(sv, exp_depth, d["ping_time"], "2m", "20s")
    close(mv["Sv"].values, exp, 1e-9, "MVBS on depth")
    np.testing.assert_array_equal(mv["depth"].values, r_left)
    # upward-looking, per-ping offsets given on their own time axis (nearest alignment)
    t_off = d["ping_time"][::10]
    off = ep.DataArray(np.linspace(100, 94, len(t_off)), ("time3",), {"time3": t_off})
    ep.consolidate.add_depth(ds, depth_offset=off, downward=False)
    idx = np.abs(d["ping_time"][:, None] - t_off[None, :]).argmin(axis=1)
    close(ds["depth"].values, off.values[idx][None, :, None] - er, 1e-14, "upward depth")
    with pytest.raises(ValueError, match="then `echodata` cannot be `None`"):
        ep.consolidate.add_depth(ds, use_platform_angles=True)


def test_add_depth_with_echodata_platform_and_beam_groups(ep, caplog):
    """tests/consolidate/test_add_depth.py:398-604: depth from the Platform vertical offsets, the
    Platform angles and the Beam-group direction vectors of an EK60 / EK80 EchoData."""
    d = ep.synth.ek60_numpy(2, 40, 300)
    ed = ep.echodata.from_ek60_arrays(d)
    ds = ep.calibrate.compute_Sv(ed)
    _, er = oc.ek60(d, "Sv")
    t2 = d["ping_time"][::4] + np.timedelta64(300, "ms")
    rng = np.random.default_rng(0)
    plat = ep.Dataset(coords={"time2": t2, "channel": ds["channel"].values})
    plat["water_level"] = (("time2",), rng.uniform(0, 1, t2.size))
    plat["vertical_offset"] = (("time2",), rng.uniform(-0.5, 0.5, t2.size))
    plat["transducer_offset_z"] = (("channel",), np.array([4.0, 6.5]))
    plat["pitch"] = (("time2",), rng.uniform(-10, 10, t2.size))
    plat["roll"] = (("time2",), rng.uniform(-10, 10, t2.size))
    ed["Platform"] = plat
    ed["Sonar"].attrs["sonar_model"] = "EK60"
    idx = np.abs(d["ping_time"][:, None] - t2[None, :]).argmin(axis=1)  # nearest time2 of every ping

    ep.consolidate.add_depth(ds, ed, use_platform_vertical_offsets=True)
    td = plat["transducer_offset_z"].values[:, None] - (plat["water_level"].values + plat["vertical_offset"].values)[None, idx]
    close(ds["depth"].values, td[:, :, None] + er, 1e-14, "platform vertical offsets")
    assert "Echodata `Platform` Vertical Offsets" in ds["depth"].attrs["history"]

    ep.consolidate.add_depth(ds, ed, use_platform_angles=True)
    sc = (np.cos(np.deg2rad(plat["pitch"].values)) * np.cos(np.deg2rad(plat["roll"].values)))[idx]
    close(ds["depth"].values, er * sc[None, :, None], 1e-14, "platform angles")
    assert "Echodata `Platform` Angles" in ds["depth"].attrs["history"]

    beam = ed["Sonar/Beam_group1"]
    beam["beam_direction_x"] = (("channel",), np.array([0.0, 0.6]))
    beam["beam_direction_y"] = (("channel",), np.array([0.0, 0.0]))
    beam["beam_direction_z"] = (("channel",), np.array([1.0, 0.8]))
    ep.consolidate.add_depth(ds, ed, use_beam_angles=True, use_platform_vertical_offsets=True, downward=False)
    close(ds["depth"].values, td[:, :, None] - er * np.array([1.0, 0.8])[:, None, None], 1e-14, "beam angles")
    assert "Echodata `Beam_group1` Angles" in ds["depth"].attrs["history"]

    # user-given values win over the group variables, with the reference's warnings (:117-126)
    import logging

    with caplog.at_level(logging.WARNING):
        ep.consolidate.add_depth(ds, ed, depth_offset=9.0, tilt=0.0, use_platform_vertical_offsets=True,
                                 use_platform_angles=True)
    assert "platform vertical offset variables will not be used" in caplog.text
    assert "beam/platform angle variables will not be used" in caplog.text
    close(ds["depth"].values, 9.0 + er, 1e-14, "explicit arguments win")
    with pytest.raises(NotImplementedError, match="both platform and beam angles"):
        ep.consolidate.add_depth(ds, ed, use_platform_angles=True, use_beam_angles=True)
    ed["Sonar"].attrs["sonar_model"] = "AZFP"
    with pytest.raises(NotImplementedError, match="not implemented yet for `AZFP`"):
        ep.consolidate.add_depth(ds, ed, use_beam_angles=True)
    with pytest.raises(ValueError, match="must contain a single dimension"):
        ep.consolidate.add_depth(ds, tilt=ep.DataArray(np.zeros((2, 2)), ("a", "b")))


# ------------------------------------------------------------------------------------ clean
def test_remove_background_noise_reference_kat(ep):
    for make, n_nan in ((kf.noise_toy, None), (kf.noise_seed1, 6)):
        Sv, er, a = make()
        ds = sv_dataset(ep, dict(Sv=Sv, echo_range=er, ping_time=kf.gen_ping_time(10, "1.6s")),
                        {"sound_absorption": np.asarray(a)})
        out = ep.clean.remove_noise(ds, ping_num=2, range_sample_num=5, SNR_threshold="0dB")  # legacy alias
        c = out["Sv_corrected"].values
        if n_nan is None:
            assert np.isnan(c[0, 0, 30]) and np.isnan(c[0, 0, 60])
        else:
            assert np.count_nonzero(np.isnan(c[0, :, :50])) == n_nan
        est = ep.clean.estimate_background_noise(ds, 2, 5)
        np.testing.assert_array_equal(est.values, out["Sv_noise"].values)
    with pytest.raises(TypeError):
        ep.clean.remove_background_noise(ds, 2, 5, SNR_threshold=3.0)
    with pytest.raises(ValueError):
        ep.clean.remove_background_noise(ds, 2, 5, background_noise_max="-125")


# ------------------------------------------------------------ goldens from the reference's own methods
def test_power_chains_match_reference_method_goldens(ep):
    """The HIP path against tests/golden/ref_chain_goldens.npz: Sv / TS / echo_range produced by the
    reference's own compute_range_EK / range_mod_TVG_EK / compute_range_AZFP / _cal_power_samples
    (executed over a named-dimension shim by oracle/gen_chain_goldens.py) -- EK60, EK80 power with a GPT
    channel, AZFP."""
    import os

    import torch

    gdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    g = dict(np.load(os.path.join(gdir, "ref_chain_goldens.npz")))
    g.update(np.load(os.path.join(gdir, "ref_seam_goldens.npz")))  # ek60seam: S = 2052, two chunk boundaries + a tail
    ops = ep.ops
    dev = lambda a, dt=torch.float64: torch.from_numpy(np.ascontiguousarray(a)).cuda().to(dt)  # noqa: E731
    for tag, sonar in (("ek60", "EK60"), ("ek80p", "EK80"), ("ek60psi", "EK60"),  # ek60psi: psi per (channel, ping)
                       ("ek60seam", "EK60")):
        gpt = dev(g[f"{tag}_is_gpt"].astype(np.uint8), torch.uint8) if sonar == "EK80" else None
        for cal in ("Sv", "TS"):
            coef = ops.power_coef_ek(
                dev(g[f"{tag}_sample_interval"]), dev(g[f"{tag}_tau"]), dev(g[f"{tag}_transmit_power"]),
                dev(g[f"{tag}_sound_speed"]), dev(g[f"{tag}_absorption"]), dev(g[f"{tag}_gain"]), dev(g[f"{tag}_sa"]),
                dev(g[f"{tag}_psi"]), dev(g[f"{tag}_frequency"]), dev(g[f"{tag}_tau_effective"]), sonar=sonar,
                cal_type=cal, gpt=gpt)
            out, rng = ops.sv_power(dev(g[f"{tag}_raw"], torch.float32), coef, cal_type=cal)
            close(out.cpu().numpy(), g[f"{tag}_{cal}"], 1e-9, f"{tag} {cal}")
            np.testing.assert_array_equal(rng.cpu().numpy(), g[f"{tag}_echo_range"])
    # a (channel, ping_time) equivalent_beam_angle through the Dataset API (calibrate_ek.py:154-162 broadcasts it)
    tag = "ek60psi"
    C, P, S = g[f"{tag}_raw"].shape
    chans = [f"ch{i}" for i in range(C)]
    t = np.datetime64("2026-05-01T00:00:00", "ns") + np.arange(P) * np.timedelta64(1, "s")
    d = ep.synth.ek60_numpy(C, P, S)
    d.update(backscatter_r=g[f"{tag}_raw"], sample_interval=g[f"{tag}_sample_interval"], channel=chans, ping_time=t,
             transmit_duration_nominal=g[f"{tag}_tau"], transmit_power=g[f"{tag}_transmit_power"],
             frequency_nominal=g[f"{tag}_frequency"])
    cp = lambda a: ep.DataArray(a, ("channel", "ping_time"), {"channel": chans, "ping_time": t})  # noqa: E731
    env = {"sound_speed": cp(g[f"{tag}_sound_speed"]), "sound_absorption": cp(g[f"{tag}_absorption"])}
    calp = {"gain_correction": cp(g[f"{tag}_gain"]), "sa_correction": cp(g[f"{tag}_sa"]),
            "equivalent_beam_angle": cp(g[f"{tag}_psi"])}
    for ed in (ep.echodata.from_ek60_arrays(d), ep.echodata.from_ek60_arrays(d).to_device()):
        for cal in ("Sv", "TS"):
            fn = ep.calibrate.compute_Sv if cal == "Sv" else ep.calibrate.compute_TS
            ds = fn(ed, env_params=env, cal_params=calp)
            close(ds[cal].values, g[f"{tag}_{cal}"], 1e-9, f"{tag} API {cal}")
            np.testing.assert_array_equal(ds["echo_range"].values, g[f"{tag}_echo_range"])
        np.testing.assert_array_equal(ds["equivalent_beam_angle"].values, g[f"{tag}_psi"])
        # ... and with compute_MVBS as the first reader of the deferred Sv: the array the FUSED kernel writes is held to
        # the same reference-executed values (and range_var_max caps the grid as on the plain route)
        ds = ep.calibrate.compute_Sv(ed, env_params=env, cal_params=calp)
        deferred = not ds["Sv"].data.materialized
        mv = ep.commongrid.compute_MVBS(ds, range_bin="0.5m", ping_time_bin="3s", range_var_max="4m")
        assert deferred and ds["Sv"].data.materialized and not ds["echo_range"].data.materialized
        close(ds["Sv"].values, g[f"{tag}_Sv"], 1e-9, f"{tag} API Sv written by compute_MVBS")
        assert mv["echo_range"].values.max() == 4.0 and np.isfinite(mv["Sv"].values).any()
        exp, _, _ = ogrid.compute_MVBS(g[f"{tag}_Sv"], g[f"{tag}_echo_range"], t, "0.5m", "3s", range_var_max="4m")
        close(mv["Sv"].values, exp, 1e-9, f"{tag} MVBS of the deferred route")
    # AZFP through the Dataset API with the golden's parameters as user env / cal params
    C, P, S = g["azfp_counts"].shape
    t = np.datetime64("2026-05-01T00:00:00", "ns") + np.arange(P) * np.timedelta64(2, "s")
    d = dict(backscatter_r=g["azfp_counts"].astype(np.float32), frequency_nominal=np.array([38e3, 125e3, 200e3])[:C],
             channel=[f"ch{i}" for i in range(C)], transmit_duration_nominal=np.tile(g["azfp_tau"][:, None], (1, P)),
             number_of_samples_per_average_bin=g["azfp_N"], digitization_rate=g["azfp_f"], lock_out_index=g["azfp_L"],
             EL=g["azfp_EL"], DS=g["azfp_DS"], TVR=g["azfp_TVR"], VTX0=g["azfp_VTX0"], Sv_offset=g["azfp_Sv_offset"],
             equivalent_beam_angle=g["azfp_equivalent_beam_angle"], temperature=np.full(P, 8.0), ping_time=t)
    ed = ep.echodata.from_azfp_arrays(d)
    env = {"salinity": 30.0, "pressure": 50.0,
           "sound_speed": ep.DataArray(np.tile(g["azfp_sound_speed"], (C, 1)), ("channel", "ping_time"),
                                       {"channel": d["channel"], "ping_time": t}),
           "sound_absorption": ep.DataArray(g["azfp_absorption"], ("channel",), {"channel": d["channel"]})}
    for cal in ("Sv", "TS"):
        fn = ep.calibrate.compute_Sv if cal == "Sv" else ep.calibrate.compute_TS
        ds = fn(ed, env_params=env)
        close(ds[cal].values, g[f"azfp_{cal}"], 1e-9, f"AZFP {cal}")
        close(ds["echo_range"].values, g[f"azfp_echo_range_{cal}"], 1e-13, f"AZFP echo_range {cal}")


@pytest.mark.parametrize("tag", ["n0", "n1", "n2"])
def test_remove_background_noise_matches_reference_function_goldens(ep, tag):
    """clean.remove_background_noise against tests/golden/ref_noise_goldens.npz = outputs of the reference's own
    estimate_background_noise / remove_background_noise (oracle/gen_noise_goldens.py)."""
    import os

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_noise_goldens.npz"))
    ping_num, rsn, nmax, snr = g[f"{tag}_args"]
    sv = g[f"{tag}_Sv"]
    C, P, S = sv.shape
    t = np.datetime64("2026-05-01T00:00:00", "ns") + np.arange(P) * np.timedelta64(1, "s")
    ds = sv_dataset(ep, dict(Sv=sv, echo_range=g[f"{tag}_echo_range"], ping_time=t),
                    {"sound_absorption": (("channel", "ping_time"), g[f"{tag}_absorption"])})
    kw = dict(background_noise_max=None if np.isnan(nmax) else f"{nmax}dB", SNR_threshold=f"{snr}dB")
    est = ep.clean.estimate_background_noise(ds, int(ping_num), int(rsn), background_noise_max=kw["background_noise_max"])
    close(est.values, g[f"{tag}_Sv_noise"], 1e-11, "estimate_background_noise")
    out = ep.clean.remove_background_noise(ds, int(ping_num), int(rsn), **kw)
    close(out["Sv_noise"].values, g[f"{tag}_Sv_noise"], 1e-11, "Sv_noise")
    close(out["Sv_corrected"].values, g[f"{tag}_Sv_corrected"], 1e-9, "Sv_corrected")
    got_rng = out["Sv_noise"].attrs["actual_range"] + out["Sv_corrected"].attrs["actual_range"]
    np.testing.assert_allclose(got_rng, g[f"{tag}_noise_attrs_range"], atol=0.011)  # rounded to 2 decimals


def test_to_device_keeps_a_host_view_of_the_parameters(ep):
    """EchoData.to_device(): the per-(channel, ping) parameters are readable on the host without a device-to-host
    copy (the calibrators decide e.g. "is the pulse the same on every ping?" from them), the view is read-only, and it
    is dropped as soon as the device tensor is modified in place; the sample planes carry no view."""
    import torch

    from echopype_amd.xr_lite import DeviceArray, host_readable

    d = ep.synth.ek60_numpy(2, 30, 64)
    ed = ep.echodata.from_ek60_arrays(d).to_device()
    beam = ed["Sonar/Beam_group1"]
    ss = beam["transmit_duration_nominal"].data
    assert isinstance(ss, DeviceArray) and host_readable(ss)
    a = np.asarray(ss)
    np.testing.assert_array_equal(a, d["transmit_duration_nominal"])
    assert not a.flags.writeable
    ss.tensor.mul_(2.0)  # in place: the host view is stale now
    assert not host_readable(ss)
    np.testing.assert_array_equal(np.asarray(ss), 2.0 * d["transmit_duration_nominal"])
    big = DeviceArray(torch.zeros(4, device="cuda"))
    assert not host_readable(big)


def test_pulse_table_parameters_of_the_output_are_looked_up_when_read(ep):
    """gain_correction / sa_correction of a device-resident file: the calibration kernel takes the Vendor_specific
    (C, K) tables themselves; the (channel, ping_time) variables of the output dataset run their look-up kernel when
    somebody reads them -- and then hold what the reference's get_vend_cal_params_power returns (cal_params.py:261-324,
    here: the oracle's restatement), pulse lengths that change from ping to ping included."""
    from oracle import calibrate as ocal

    from echopype_amd import _lib

    d = ep.synth.ek60_numpy(3, 50, 64, vary_tau=True)
    ed = ep.echodata.from_ek60_arrays(d).to_device()
    with _lib.launch_trace() as tr:
        ds = ep.calibrate.compute_Sv(ed)
        ds["Sv"].values
    assert "pulse_table_lookup_kernel" not in tr.kernels and "power_coef_ek_kernel" in tr.kernels
    with _lib.launch_trace() as tr:
        g, sa = ds["gain_correction"], ds["sa_correction"]
        assert g.dims == ("channel", "ping_time") and g.shape == (3, 50)
        gv, sav = g.values, sa.values
    assert tr.kernels.count("pulse_table_lookup_kernel") == 2
    np.testing.assert_array_equal(gv, ocal.vend_cal_params_power(d["transmit_duration_nominal"], d["pulse_length"],
                                                                 d["gain_correction"]))
    np.testing.assert_array_equal(sav, ocal.vend_cal_params_power(d["transmit_duration_nominal"], d["pulse_length"],
                                                                  d["sa_correction"]))
    # ... and the host route (no device-resident parameters) holds the same numbers
    ds_h = ep.calibrate.compute_Sv(ep.echodata.from_ek60_arrays(d))
    np.testing.assert_array_equal(ds_h["gain_correction"].values, gv)
    np.testing.assert_array_equal(ds_h["Sv"].values, ds["Sv"].values)


# ---- echo_range left lazy by compute_Sv on power samples -------------------------------------------------------------
def _lazy_case(ep, S=1000, dtype="float64", P=240):
    d = ep.synth.ek60_numpy(3, P, S, ss_every=7)
    d["backscatter_r"][1, 5, S - 60:] = np.nan      # a NaN-padded ping: echo_range is NaN there (range.py:143-148)
    d["backscatter_r"][0, 11, 3] = np.nan
    return ep.echodata.from_ek60_arrays(d), dtype


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_compute_Sv_leaves_echo_range_lazy_and_identical(dtype):
    """compute_Sv on power samples does not write echo_range (8 of the pass's 20 B/sample): the Dataset variable is a
    LazyDeviceArray carrying shape, dtype and the {nanmin, nanmax, NaN count} by-product; reading it produces exactly
    the array K1 writes (bit-identical, NaN mask of padded samples included), and the statistics are those of that
    array."""
    import torch
    import echopype_amd as ep
    from echopype_amd import ops
    from echopype_amd.xr_lite import LazyDeviceArray

    ed, _ = _lazy_case(ep)
    ds = ep.calibrate.compute_Sv(ed, dtype=dtype)
    rng = ds["echo_range"].data
    assert isinstance(rng, LazyDeviceArray) and not rng.materialized
    assert rng.shape == ds["Sv"].shape and rng.dtype == np.dtype(dtype)
    lo, hi, nn = rng.cached_stats()
    assert not rng.materialized
    # the eager pass, through the C ABI
    cal = ep.calibrate.api.CALIBRATOR["EK60"](ed, None, None, None, dtype=dtype)
    raw, coef, flags, _ = cal._power_inputs("Sv")
    sv_e, rg_e, st_e = ops.sv_power(raw, coef, flags=flags, dtype=getattr(torch, dtype), want_range_stats=True)
    got = ds["echo_range"].values
    assert rng.materialized
    np.testing.assert_array_equal(got, rg_e.cpu().numpy())
    np.testing.assert_array_equal(ds["Sv"].values, sv_e.cpu().numpy())
    assert (lo, hi, nn) == tuple(st_e.cpu().tolist()[:2]) + (int(st_e[2]),)
    raw_nan = np.isnan(ed["Sonar/Beam_group1"]["backscatter_r"].values)
    assert np.isnan(got[1, 5, -60:]).all() and np.isnan(got[0, 11, 3]) and nn == int(raw_nan.sum())
    np.testing.assert_array_equal(np.isnan(got), raw_nan)
    assert rng.cached_stats() == (lo, hi, nn)          # still valid after the array has been written


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["float64", "float32"])
@pytest.mark.parametrize("closed", ["left", "right"])
def test_compute_MVBS_bins_a_lazy_echo_range_through_its_coefficient_rows(dtype, closed):
    """compute_MVBS on the Dataset straight from compute_Sv never writes echo_range (the binning kernel evaluates the
    coefficient rows) and gives the MVBS of the same call on the materialised array (1e-13); skipna=False needs the
    NaN mask of the array and takes it."""
    import echopype_amd as ep

    ed, _ = _lazy_case(ep)
    ds = ep.calibrate.compute_Sv(ed, dtype=dtype)
    lazy = ds["echo_range"].data
    a = ep.commongrid.compute_MVBS(ds, range_bin="2m", ping_time_bin="10s", closed=closed)
    assert not lazy.materialized
    ds2 = ep.calibrate.compute_Sv(ed, dtype=dtype)
    ds2["echo_range"].values                                   # written: the array path from here on
    ds2["echo_range"] = ep.xr_lite.DataArray(ep.DeviceArray(ds2["echo_range"].data.tensor), ds2["echo_range"].dims)
    b = ep.commongrid.compute_MVBS(ds2, range_bin="2m", ping_time_bin="10s", closed=closed)
    # (same bins, same members; the two kernel instantiations add a bin's members in different orders: last-bit noise)
    np.testing.assert_array_equal(np.isnan(a["Sv"].values), np.isnan(b["Sv"].values))
    np.testing.assert_allclose(a["Sv"].values, b["Sv"].values, rtol=1e-12 if dtype == "float64" else 1e-5, atol=1e-12 if dtype == "float64" else 1e-4)
    np.testing.assert_array_equal(a["echo_range"].values, b["echo_range"].values)
    assert np.isfinite(a["Sv"].values).any()
    c = ep.commongrid.compute_MVBS(ds, range_bin="2m", ping_time_bin="10s", closed=closed, skipna=False)
    assert lazy.materialized
    d = ep.commongrid.compute_MVBS(ds2, range_bin="2m", ping_time_bin="10s", closed=closed, skipna=False)
    np.testing.assert_allclose(c["Sv"].values, d["Sv"].values, rtol=1e-12 if dtype == "float64" else 1e-5, atol=1e-12 if dtype == "float64" else 1e-4)


@pytest.mark.gpu
def test_lazy_echo_range_edge_cases():
    """An odd number of samples per ping (K1's one-sample-per-lane path) keeps the eager array; backscatter modified
    in place between compute_Sv and the first read of a lazy echo_range raises instead of returning a wrong mask;
    remove_background_noise on the lazy dataset leaves it lazy and returns what it returns on the array."""
    import torch
    import echopype_amd as ep
    from echopype_amd.xr_lite import LazyDeviceArray

    ed, _ = _lazy_case(ep, S=999)
    ds = ep.calibrate.compute_Sv(ed)
    assert not isinstance(ds["echo_range"].data, LazyDeviceArray)
    ed, _ = _lazy_case(ep)
    ed.to_device()
    ds = ep.calibrate.compute_Sv(ed)
    assert isinstance(ds["echo_range"].data, LazyDeviceArray)
    ed["Sonar/Beam_group1"]["backscatter_r"].data.tensor[0, 0, 0] = 1.0
    with pytest.raises(RuntimeError, match="modified in place"):
        ds["echo_range"].values
    for dtype in ("float64", "float32"):
        # remove_background_noise / estimate_background_noise on the lazy dataset: coefficient rows + the NaN pattern of
        # the raw samples instead of the array (never written) == the same calls on the array
        ed, _ = _lazy_case(ep)
        ed["Sonar/Beam_group1"]["backscatter_r"].values[2, 17, :] = np.nan       # a dropped ping: NaN from sample 0 on
        ds = ep.calibrate.compute_Sv(ed, dtype=dtype)
        est = ep.clean.estimate_background_noise(ds, 20, 50)
        out = ep.clean.remove_background_noise(ds, 20, 50)
        assert not ds["echo_range"].data.materialized
        ds2 = ep.calibrate.compute_Sv(ed, dtype=dtype)
        ds2["echo_range"] = ep.xr_lite.DataArray(ep.DeviceArray(ds2["echo_range"].data.tensor), ds2["echo_range"].dims)
        est2 = ep.clean.estimate_background_noise(ds2, 20, 50)
        out2 = ep.clean.remove_background_noise(ds2, 20, 50)
        # (equal up to the last-bit noise of the estimate's LDS atomics, which moves whole ping blocks of Sv_noise)
        ntol = dict(rtol=1e-13, atol=1e-12) if dtype == "float64" else dict(rtol=2e-6, atol=1e-4)
        for a, b in ((est, est2), (out["Sv_noise"], out2["Sv_noise"])):
            np.testing.assert_array_equal(np.isnan(a.values), np.isnan(b.values))
            np.testing.assert_allclose(a.values, b.values, **ntol)
        ca, cb = out["Sv_corrected"].values, out2["Sv_corrected"].values
        both = np.isfinite(ca) & np.isfinite(cb)
        assert (np.isnan(ca) != np.isnan(cb)).mean() < 2e-3
        np.testing.assert_allclose(ca[both], cb[both], rtol=1e-9 if dtype == "float64" else 1e-3, atol=1e-9 if dtype == "float64" else 1e-4)
        sn = out["Sv_noise"].values
        assert np.isnan(sn[2, 17]).all() and np.isnan(sn[1, 5, -60:]).all() and np.isfinite(sn[0, 0, :3]).all()
        assert np.isnan(out["Sv"].values[0, 0, :2]).all()   # (R' <= 0: Sv is NaN there, echo_range and Sv_noise are not)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_ek80_bb_echo_range_lazy_and_mvbs_through_rows(ep, dtype):
    """EK80 broadband compute_Sv (the LDS-FFT form) leaves echo_range lazy too: statistics from the sample pass,
    the array from epa_range_complex on first read (bit-identical to the eager range_out, NaN where sector 0 is),
    compute_MVBS through the coefficient rows == compute_MVBS on the array; the same for CW complex samples (the
    streaming kernel's by-product), echo_range bit-identical to the plain epa_sv_complex call."""
    import torch
    from echopype_amd import ops
    from echopype_amd.xr_lite import LazyDeviceArray

    d, filt = _ek80(ep, "BB", C=2, P=40, S=1200, mixed_nan=True)
    ed = ep.echodata.from_ek80_arrays(d, filt)
    ds = ep.calibrate.compute_Sv(ed, waveform_mode="BB", encode_mode="complex", dtype=dtype)
    lazy = ds["echo_range"].data
    assert isinstance(lazy, LazyDeviceArray) and not lazy.materialized
    a = ep.commongrid.compute_MVBS(ds, range_bin="0.5m", ping_time_bin="5s")
    assert not lazy.materialized
    cal = ep.calibrate.api.CALIBRATOR["EK80"](ed, None, None, "BB", "complex", dtype=dtype)
    k, _ = cal._complex_inputs("Sv")
    eager = ops.sv_complex(k["re"], k["im"], k["ccoef"], replica=k["replica"], replica_off=k["replica_off"],
                           max_taps=k["max_taps"], dtype=getattr(torch, dtype), want_range_stats=True)
    stats = lazy.cached_stats()
    got = ds["echo_range"].values
    np.testing.assert_array_equal(got, eager["echo_range"].cpu().numpy())
    np.testing.assert_array_equal(np.isnan(got), np.isnan(d["backscatter_r"][..., 0]))
    lo, hi, nn = eager["range_stats"].cpu().tolist()
    assert stats == (lo, hi, int(nn)) and nn == np.isnan(got).sum() and hi == np.nanmax(got)
    ds2 = ep.calibrate.compute_Sv(ed, waveform_mode="BB", encode_mode="complex", dtype=dtype)
    ds2["echo_range"] = ep.xr_lite.DataArray(ep.DeviceArray(ds2["echo_range"].data.tensor), ds2["echo_range"].dims)
    b = ep.commongrid.compute_MVBS(ds2, range_bin="0.5m", ping_time_bin="5s")
    np.testing.assert_array_equal(np.isnan(a["Sv"].values), np.isnan(b["Sv"].values))
    np.testing.assert_allclose(a["Sv"].values, b["Sv"].values, rtol=1e-12 if dtype == "float64" else 1e-5, atol=1e-12 if dtype == "float64" else 1e-4)
    assert np.isfinite(a["Sv"].values).any()
    # CW complex samples: the streaming kernel leaves the statistics too (epa_sv_complex_cw_stats)
    dcw, filt = _ek80(ep, "CW", C=2, P=30, S=2100, mixed_nan=True)
    edcw = ep.echodata.from_ek80_arrays(dcw, filt)
    dscw = ep.calibrate.compute_Sv(edcw, waveform_mode="CW", encode_mode="complex", dtype=dtype)
    lz = dscw["echo_range"].data
    assert isinstance(lz, LazyDeviceArray) and not lz.materialized
    calcw = ep.calibrate.api.CALIBRATOR["EK80"](edcw, None, None, "CW", "complex", dtype=dtype)
    kc, _ = calcw._complex_inputs("Sv")
    eg = ops.sv_complex(kc["re"], kc["im"], kc["ccoef"], dtype=getattr(torch, dtype))
    st = lz.cached_stats()
    gr = dscw["echo_range"].values
    np.testing.assert_array_equal(gr, eg["echo_range"].cpu().numpy())
    # (two instantiations of the kernel: the compiler contracts the dB sum differently, last-bit differences)
    np.testing.assert_array_equal(np.isnan(dscw["Sv"].values), np.isnan(eg["out"].cpu().numpy()))
    np.testing.assert_allclose(dscw["Sv"].values, eg["out"].cpu().numpy(), rtol=1e-14 if dtype == "float64" else 1e-6)
    assert st == (np.nanmin(gr), np.nanmax(gr), int(np.isnan(gr).sum())) and st[2] > 0


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_add_depth_from_a_lazy_echo_range_and_its_statistics(dtype):
    """add_depth on the Dataset straight from compute_Sv evaluates the coefficient rows (echo_range stays unwritten)
    and gives bit-identical depth to add_depth on the array; the depth variable carries {nanmin, nanmax, NaN count},
    so compute_MVBS(range_var="depth") does not sweep it again -- same MVBS either way."""
    import echopype_amd as ep

    ed, _ = _lazy_case(ep)
    tilt = ep.xr_lite.DataArray(np.linspace(0.0, 20.0, 240), ("ping_time",),
                                coords={"ping_time": ed["Sonar/Beam_group1"]["ping_time"].values})
    ds = ep.calibrate.compute_Sv(ed, dtype=dtype)
    out = ep.consolidate.add_depth(ds, depth_offset=4.5, tilt=tilt)
    assert not ds["echo_range"].data.materialized
    ds2 = ep.calibrate.compute_Sv(ed, dtype=dtype)
    ds2["echo_range"] = ep.xr_lite.DataArray(ep.DeviceArray(ds2["echo_range"].data.tensor), ds2["echo_range"].dims)
    out2 = ep.consolidate.add_depth(ds2, depth_offset=4.5, tilt=tilt)
    got = out["depth"].values
    np.testing.assert_array_equal(got, out2["depth"].values)
    er = ds2["echo_range"].values
    exp = 4.5 + np.cos(np.deg2rad(tilt.values))[None, :, None].astype(dtype) * er   # consolidate/api.py:226
    np.testing.assert_allclose(got, exp, rtol=1e-14 if dtype == "float64" else 1e-6)
    for o in (out, out2):
        assert o["depth"].data.cached_stats() == (np.nanmin(got), np.nanmax(got), int(np.isnan(got).sum()))
    a = ep.commongrid.compute_MVBS(out, range_var="depth", range_bin="2m", ping_time_bin="10s")
    b = ep.commongrid.compute_MVBS(out2, range_var="depth", range_bin="2m", ping_time_bin="10s")
    np.testing.assert_array_equal(np.isnan(a["Sv"].values), np.isnan(b["Sv"].values))
    # (the two runs add a bin's members in different orders -- LDS atomics: last-bit noise)
    np.testing.assert_allclose(a["Sv"].values, b["Sv"].values, rtol=1e-12 if dtype == "float64" else 1e-5,
                               atol=1e-12 if dtype == "float64" else 1e-4)
    assert np.isfinite(a["Sv"].values).any()


# ---- Sv left to its first reader by compute_Sv on power samples; compute_MVBS writes it next to the bins ---------------
@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["float64"])
def test_deferred_sv_is_written_by_compute_MVBS_and_equals_the_eager_calls(dtype, monkeypatch):
    """compute_Sv on EK power samples returns Sv as a LazyDeviceArray; compute_MVBS right after runs ONE pass over the raw
    samples (epa_sv_mvbs_fused) that writes the Sv array and the bins.  Same dataset as with EPA_DEFER_SV=0 (K1, then the
    binning kernel on the Sv array): Sv bit for bit, MVBS up to the order of a bin's additions, the echo_range statistics,
    the NaN-coordinate warning."""
    import logging

    import echopype_amd as ep
    from echopype_amd.xr_lite import LazyDeviceArray

    ed, _ = _lazy_case(ep)
    monkeypatch.setenv("EPA_DEFER_SV", "0")
    ds_e = ep.calibrate.compute_Sv(ed, dtype=dtype)
    assert not isinstance(ds_e["Sv"].data, LazyDeviceArray)
    mv_e = ep.commongrid.compute_MVBS(ds_e, range_bin="2m", ping_time_bin="10s")
    monkeypatch.delenv("EPA_DEFER_SV")

    ds = ep.calibrate.compute_Sv(ed, dtype=dtype)
    sv, rng = ds["Sv"].data, ds["echo_range"].data
    assert isinstance(sv, LazyDeviceArray) and not sv.materialized and sv.source is not None
    assert ds["Sv"].shape == ds_e["Sv"].shape and ds["Sv"].dtype == np.dtype(dtype)
    assert ds["Sv"].attrs == ds_e["Sv"].attrs and list(ds.data_vars) == list(ds_e.data_vars)
    records = []
    handler = logging.Handler()
    handler.emit = records.append
    logging.getLogger().addHandler(handler)
    from echopype_amd.xr_lite import DeferredDataset
    try:
        mv = ep.commongrid.compute_MVBS(ds, range_bin="2m", ping_time_bin="10s")
        # the call launched the kernel and returned: the grid np.arange(0, nanmax(echo_range) + bin, bin) needs a number
        # that kernel produces, so the dataset is assembled (and the NaN-coordinate warning logged) on first use
        assert isinstance(mv, DeferredDataset) and not mv.resolved and not records
        assert sv.materialized and sv.source is None  # (the Sv array is the kernel's output buffer already)
        assert set(mv.sizes) == {"ping_time", "channel", "echo_range"} and mv.resolved
    finally:
        logging.getLogger().removeHandler(handler)
    assert any("coordinate array contain NaNs" in r.getMessage() for r in records)   # NaN-padded pings in _lazy_case
    assert sv.materialized and sv.source is None and not rng.materialized
    np.testing.assert_array_equal(ds["Sv"].values, ds_e["Sv"].values)
    assert rng.cached_stats() == ds_e["echo_range"].data.cached_stats()
    np.testing.assert_array_equal(mv["echo_range"].values, mv_e["echo_range"].values)
    np.testing.assert_array_equal(mv["ping_time"].values, mv_e["ping_time"].values)
    np.testing.assert_array_equal(np.isnan(mv["Sv"].values), np.isnan(mv_e["Sv"].values))
    tol = dict(rtol=1e-12, atol=1e-12) if dtype == "float64" else dict(rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(mv["Sv"].values, mv_e["Sv"].values, **tol)
    assert mv["Sv"].attrs == mv_e["Sv"].attrs
    # a second grid on the same dataset: the plain route on the Sv array now there, echo_range still only its rows
    mv2 = ep.commongrid.compute_MVBS(ds, range_bin="5m", ping_time_bin="20s")
    mv2_e = ep.commongrid.compute_MVBS(ds_e, range_bin="5m", ping_time_bin="20s")
    assert not rng.materialized
    np.testing.assert_allclose(mv2["Sv"].values, mv2_e["Sv"].values, **tol)


@pytest.mark.gpu
def test_deferred_sv_other_first_readers(monkeypatch):
    """Whoever reads a deferred Sv first gets the array K1 writes: .values, remove_background_noise, a right-closed or
    skipna=False compute_MVBS (the plain route), the echo_range statistics asked for before anything else.  Raw samples
    modified in place in between: an error instead of values of the wrong samples."""
    import echopype_amd as ep
    from echopype_amd.xr_lite import LazyDeviceArray

    ed, _ = _lazy_case(ep)
    monkeypatch.setenv("EPA_DEFER_SV", "0")
    ds_e = ep.calibrate.compute_Sv(ed)
    st_e = ds_e["echo_range"].data.cached_stats()
    mv_r = ep.commongrid.compute_MVBS(ds_e, range_bin="2m", ping_time_bin="10s", closed="right")
    clean_e = ep.clean.remove_background_noise(ep.calibrate.compute_Sv(ed), ping_num=20, range_sample_num=50)
    monkeypatch.delenv("EPA_DEFER_SV")

    ds = ep.calibrate.compute_Sv(ed)
    assert not ds["Sv"].data.materialized
    assert ds["echo_range"].data.cached_stats() == st_e      # the statistics come with the Sv pass: it ran now
    assert ds["Sv"].data.materialized and not ds["echo_range"].data.materialized
    np.testing.assert_array_equal(ds["Sv"].values, ds_e["Sv"].values)

    ds = ep.calibrate.compute_Sv(ed)
    np.testing.assert_array_equal(ds["Sv"].values, ds_e["Sv"].values)         # .values first
    ds = ep.calibrate.compute_Sv(ed)
    mv = ep.commongrid.compute_MVBS(ds, range_bin="2m", ping_time_bin="10s", closed="right")
    np.testing.assert_allclose(mv["Sv"].values, mv_r["Sv"].values, rtol=1e-12, atol=1e-12)
    assert ds["Sv"].data.materialized
    # remove_background_noise first: two passes over the raw samples (Sv + estimate, then Sv_noise / Sv_corrected) instead
    # of K1 and two sweeps of the Sv array -- the estimate sums through LDS atomics either way (last-bit noise), and the
    # odd sample sits exactly on the SNR threshold
    ds = ep.calibrate.compute_Sv(ed)
    clean = ep.clean.remove_background_noise(ds, ping_num=20, range_sample_num=50)
    assert ds["Sv"].data.materialized and not ds["echo_range"].data.materialized
    assert ds["echo_range"].data.cached_stats() == st_e
    np.testing.assert_array_equal(clean["Sv"].values, clean_e["Sv"].values)
    np.testing.assert_array_equal(np.isnan(clean["Sv_noise"].values), np.isnan(clean_e["Sv_noise"].values))
    np.testing.assert_allclose(clean["Sv_noise"].values, clean_e["Sv_noise"].values, rtol=1e-13, atol=1e-12)
    ca, cb = clean["Sv_corrected"].values, clean_e["Sv_corrected"].values
    both = np.isfinite(ca) & np.isfinite(cb)
    assert both.any() and (np.isnan(ca) != np.isnan(cb)).mean() < 2e-3
    np.testing.assert_allclose(ca[both], cb[both], rtol=1e-9, atol=1e-9)
    for name in ("Sv_noise", "Sv_corrected"):
        assert set(clean[name].attrs) == set(clean_e[name].attrs)
        np.testing.assert_allclose(clean[name].attrs["actual_range"], clean_e[name].attrs["actual_range"], atol=0.011)
    mv = ep.commongrid.compute_MVBS(clean, range_bin="2m", ping_time_bin="10s")     # (the rows route: range still lazy)
    assert not ds["echo_range"].data.materialized and np.isfinite(mv["Sv"].values).any()
    # a copy of the dataset shares the deferred array: written once, seen by both
    ds = ep.calibrate.compute_Sv(ed)
    cp = ds.copy()
    ep.commongrid.compute_MVBS(cp, range_bin="2m", ping_time_bin="10s")
    assert ds["Sv"].data.materialized and ds["Sv"].data is cp["Sv"].data
    # TS is not deferred (nothing downstream fuses with it), nor is a float32 Sv (its MVBS bins on the float32 range)
    assert not isinstance(ep.calibrate.compute_TS(ed)["TS"].data, LazyDeviceArray)
    assert not isinstance(ep.calibrate.compute_Sv(ed, dtype="float32")["Sv"].data, LazyDeviceArray)
    # raw samples written to between compute_Sv and the first read
    ed2, _ = _lazy_case(ep)
    ed2 = ed2.to_device()
    ds = ep.calibrate.compute_Sv(ed2)
    ed2["Sonar/Beam_group1"]["backscatter_r"].data.tensor.add_(1.0)
    with pytest.raises(RuntimeError, match="modified in place"):
        ep.commongrid.compute_MVBS(ds, range_bin="2m", ping_time_bin="10s")


def test_swap_dims_channel_frequency_then_MVBS_and_assign_actual_range(ep):
    """The reference's compute_MVBS tests on a dataset whose first dimension is frequency_nominal
    (tests/commongrid/test_commongrid_api.py:261-276) and its post-computation actual_range (:560-577)."""
    from echopype_amd.commongrid.utils import assign_actual_range

    d = ep.synth.ek60_numpy(3, 90, 400)
    ds = ep.calibrate.compute_Sv(ep.echodata.from_ek60_arrays(d))
    mv = ep.commongrid.compute_MVBS(ds, range_bin="5m", ping_time_bin="20s")
    sw = ep.consolidate.swap_dims_channel_frequency(ds)
    assert tuple(sw["Sv"].dims) == ("frequency_nominal", "ping_time", "range_sample") and "channel" not in sw.coords
    np.testing.assert_array_equal(sw["frequency_nominal"].values, ds["frequency_nominal"].values)
    np.testing.assert_array_equal(sw["channel"].values, ds["channel"].values)
    assert tuple(sw["channel"].dims) == ("frequency_nominal",)
    mvs = ep.commongrid.compute_MVBS(sw, range_bin="5m", ping_time_bin="20s")
    assert tuple(mvs["Sv"].dims) == ("frequency_nominal", "ping_time", "echo_range")
    np.testing.assert_array_equal(mvs["ping_time"].values, mv["ping_time"].values)
    np.testing.assert_allclose(mvs["Sv"].values, mv["Sv"].values, rtol=1e-12, atol=1e-12)
    np.testing.assert_array_equal(mvs["channel"].values, ds["channel"].values)
    dup = dict(d)
    dup["frequency_nominal"] = np.array([38000.0, 38000.0, 120000.0])
    with pytest.raises(ValueError, match="Duplicated transducer nominal frequencies"):
        ep.consolidate.swap_dims_channel_frequency(ep.calibrate.compute_Sv(ep.echodata.from_ek60_arrays(dup)))
    out = assign_actual_range(mv)
    v = mv["Sv"].values
    assert out.attrs["actual_range"] == [round(float(np.nanmin(v)), 2), round(float(np.nanmax(v)), 2)]
    assert "actual_range" not in mv.attrs


def test_ek80_power_samples_through_the_api_match_the_reference_golden(ep):
    """EK80 CW POWER samples (encode_mode="power", one GPT channel: calibrate_ek.py:79-206 with the EK80 flag) through the
    Dataset API -- an EchoData whose power beam group is found through Sonar.waveform_encode_descr -- against the
    reference-executed `ek80p_*` golden; Sv both as K1 writes it and as compute_MVBS writes it on the deferred route."""
    import os

    from echopype_amd.echodata import EchoData
    from echopype_amd.xr_lite import Dataset

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_chain_goldens.npz"))
    tag = "ek80p"
    raw = g[f"{tag}_raw"]
    C, P, S = raw.shape
    chans = [f"ch{i}" for i in range(C)]
    t = np.datetime64("2026-05-01T00:00:00", "ns") + np.arange(P) * np.timedelta64(1, "s")
    cpd = ("channel", "ping_time")
    beam = Dataset(coords={"channel": chans, "ping_time": t, "range_sample": np.arange(S)})
    beam["backscatter_r"] = (("channel", "ping_time", "range_sample"), raw)
    beam["sample_interval"] = (cpd, g[f"{tag}_sample_interval"])
    beam["transmit_duration_nominal"] = (cpd, g[f"{tag}_tau"])
    beam["transmit_power"] = (cpd, g[f"{tag}_transmit_power"])
    beam["frequency_nominal"] = (("channel",), g[f"{tag}_frequency"])
    sonar = Dataset(coords={"beam_group": ["Beam_group1"]})
    sonar["waveform_encode_descr"] = (("beam_group",), np.array(["power"]))
    vend = Dataset(coords={"channel": chans})
    vend["transceiver_type"] = (("channel",), np.where(g[f"{tag}_is_gpt"], "GPT", "WBT"))
    env_g = Dataset(coords={"channel": chans, "time1": t[:1]})   # what the EK80 converter always writes (set_groups_ek80.py)
    for name, v in (("temperature", 8.0), ("salinity", 33.0), ("depth", 5.0), ("acidity", 8.0), ("sound_speed_indicative", 1490.0)):
        env_g[name] = (("time1",), np.array([v]))
    ed = EchoData("EK80", {"Sonar": sonar, "Sonar/Beam_group1": beam, "Vendor_specific": vend, "Environment": env_g},
                  source_file="synthetic_ek80_power.raw")
    cp = lambda a: ep.DataArray(a, cpd, {"channel": chans, "ping_time": t})  # noqa: E731
    env = {"sound_speed": cp(g[f"{tag}_sound_speed"]), "sound_absorption": cp(g[f"{tag}_absorption"])}
    calp = {"gain_correction": cp(g[f"{tag}_gain"]), "sa_correction": cp(g[f"{tag}_sa"]),
            "equivalent_beam_angle": ep.DataArray(g[f"{tag}_psi"], ("channel",), {"channel": chans})}
    kw = dict(env_params=env, cal_params=calp, waveform_mode="CW", encode_mode="power")
    for cal in ("Sv", "TS"):
        fn = ep.calibrate.compute_Sv if cal == "Sv" else ep.calibrate.compute_TS
        ds = fn(ed, **kw)
        close(ds[cal].values, g[f"{tag}_{cal}"], 1e-9, f"{tag} API {cal}")
        np.testing.assert_array_equal(ds["echo_range"].values, g[f"{tag}_echo_range"])
    np.testing.assert_allclose(ep.calibrate.compute_Sv(ed, **kw)["tau_effective"].values, g[f"{tag}_tau_effective"], rtol=1e-15)
    ds = ep.calibrate.compute_Sv(ed, **kw)
    deferred = not ds["Sv"].data.materialized
    mv = ep.commongrid.compute_MVBS(ds, range_bin="0.5m", ping_time_bin="2s")
    assert deferred and ds["Sv"].data.materialized
    close(ds["Sv"].values, g[f"{tag}_Sv"], 1e-9, f"{tag} API Sv written by compute_MVBS")
    exp, _, _ = ogrid.compute_MVBS(g[f"{tag}_Sv"], g[f"{tag}_echo_range"], t, "0.5m", "2s")
    close(mv["Sv"].values, exp, 1e-9, f"{tag} MVBS of the deferred route")
