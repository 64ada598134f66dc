"""Host-side logic of the drop-in (parameter assembly, bin edges, containers) -- CPU only.
Checked against goldens from the reference's leaf functions, the oracle, pandas, and the
reference's synthetic unit tests (restated)."""
import warnings

import numpy as np
import pandas as pd
import pytest

import kat_fixtures as kf
from oracle import calibrate as ocal

from echopype_amd import echodata as ed_mod
from echopype_amd import synth
from echopype_amd.calibrate import cal_params, env_params
from echopype_amd.calibrate import ek80_complex as ek
from echopype_amd.calibrate.calibrate_azfp import CalibrateAZFP
from echopype_amd.clean.utils import extract_dB
from echopype_amd.commongrid import utils as gu
from echopype_amd.utils import uwa
from echopype_amd.xr_lite import DataArray, Dataset


def test_uwa_matches_reference_goldens(leaf_goldens):
    g = leaf_goldens
    for src in ("AM", "FG", "AZFP"):
        got = [uwa.calc_absorption(f, T, S, P, pH, formula_source=src) for f, T, S, P, pH in g["uwa_pts"]]
        np.testing.assert_allclose(got, g[f"uwa_abs_{src}"], rtol=1e-14)
    for src in ("Mackenzie", "AZFP"):
        got = [uwa.calc_sound_speed(T, S, P, formula_source=src) for T, S, P in g["uwa_ss_pts"]]
        np.testing.assert_allclose(got, g[f"uwa_ss_{src}"], rtol=1e-15)
    np.testing.assert_allclose(uwa.calc_absorption(g["uwa_fvec"], 10, 35, 10, 8, formula_source="FG"),
                               g["uwa_abs_FG_vec"], rtol=1e-14)
    with pytest.raises(ValueError):
        uwa.calc_absorption(38e3, formula_source="XX")


@pytest.mark.parametrize("i", [0, 1, 2])
def test_transmit_replica_matches_reference_goldens(leaf_goldens, i):
    g = leaf_goldens
    fs, tau, slope, f0, f1 = g["chan_params"][i]
    for drop in (False, True):
        y, _ = ek.tapered_chirp(fs, tau, slope, f0, f1, drop_last_hanning_zero=drop)
        np.testing.assert_allclose(y, g[f"chirp{i}_drop{int(drop)}"], rtol=1e-13, atol=1e-15)
    y, _ = ek.tapered_chirp(fs, tau, slope, f0, f1)
    ytx, t = ek.filter_decimate_chirp(dict(wbt_fil=g["wbt_fil"], wbt_decifac=6, pc_fil=g["pc_fil"], pc_decifac=2), y, fs)
    np.testing.assert_allclose(ytx, g[f"replica{i}"], rtol=1e-12, atol=1e-15)
    mode = "CW" if f0 == f1 else "BB"
    te = ek.get_tau_effective({"c": ytx}, {"c": 1 / np.diff(t[:2])}, mode)
    np.testing.assert_allclose(te.values, g[f"tau_eff{i}"], rtol=1e-12)
    np.testing.assert_allclose(ek.get_norm_fac({"c": ytx}).values, g[f"norm_fac{i}"], rtol=1e-13)


def test_filter_coeff_extraction_drops_nan_padding():
    # tests/calibrate/test_ek80_complex.py:15-75 restated: NaN-padded coefficient arrays
    d = synth.ek80_numpy(2, 2, 32)
    filt = synth.ek80_filters()
    e = ed_mod.from_ek80_arrays(d, filt)
    coeff = ek.get_filter_coeff(e["Vendor_specific"])
    for ch in e["Vendor_specific"]["channel"].values:
        np.testing.assert_array_equal(coeff[ch]["wbt_fil"], filt["wbt_fil"].astype(np.complex128))
        np.testing.assert_array_equal(coeff[ch]["pc_fil"], filt["pc_fil"].astype(np.complex128))
        assert coeff[ch]["wbt_decifac"] == 6 and coeff[ch]["pc_decifac"] == 2
    assert ek.get_vend_filter_EK80(Dataset(coords={"channel": ["a"]}), "a", "WBT", "coeff") is None


@pytest.mark.parametrize("tau,expected", kf.PULSE_CASES)
@pytest.mark.parametrize("swap", [False, True])
def test_get_vend_cal_params_power(tau, expected, swap):
    # tests/calibrate/test_cal_params.py:751-868 (incl. the channel-order-differs cases)
    vend = Dataset(coords={"channel": ["chA", "chB"], "pulse_length_bin": np.arange(4)})
    vend["pulse_length"] = (("channel", "pulse_length_bin"), kf.PULSE_TABLE["pulse_length"])
    vend["sa_correction"] = (("channel", "pulse_length_bin"), kf.PULSE_TABLE["table"])
    order = [1, 0] if swap else [0, 1]
    beam = Dataset(coords={"channel": [["chA", "chB"][i] for i in order], "ping_time": np.arange(4)})
    beam["transmit_duration_nominal"] = (("channel", "ping_time"), tau[order])
    got = cal_params.get_vend_cal_params_power(beam, vend, "sa_correction")
    np.testing.assert_array_equal(got.values, expected[order])
    with pytest.raises(ValueError, match="Unknown parameter"):
        cal_params.get_vend_cal_params_power(beam, vend, "gain")
    with pytest.raises(ValueError, match="does not exist in the Vendor_specific group"):
        cal_params.get_vend_cal_params_power(beam, vend, "gain_correction")


def test_harmonize_env_param_time():
    # tests/calibrate/test_env_params.py:33-126
    t1 = np.array(["2017-06-20T01:00:00", "2017-06-20T01:00:30", "2017-06-20T01:01:00"], "datetime64[ns]")
    p = DataArray(np.array([0.0, 1, 2]), ("time1",), {"time1": t1}, name="p")
    assert env_params.harmonize_env_param_time(5.0) == 5.0
    with pytest.raises(ValueError):
        env_params.harmonize_env_param_time(p, ping_time=None)
    same = env_params.harmonize_env_param_time(p, ping_time=t1)
    np.testing.assert_array_equal(same.values, p.values)
    q = np.array(["2017-06-20T01:00:15"], "datetime64[ns]")
    assert env_params.harmonize_env_param_time(p, ping_time=q).values[0] == 0.5
    t_long = np.arange("2017-06-20T01:00:00", "2017-06-22T01:00:31", np.timedelta64(30, "s"), dtype="datetime64[ns]")
    p2 = DataArray(np.arange(len(t_long), dtype=float), ("time1",), {"time1": t_long})
    q2 = np.array(["2017-06-20T01:00:15", "2017-06-21T01:00:15"], "datetime64[ns]")
    np.testing.assert_array_equal(env_params.harmonize_env_param_time(p2, ping_time=q2).values, [0.5, 2880.5])
    one = DataArray(np.array([[7.0], [8.0]]), ("channel", "time1"), {"time1": t1[:1]})
    assert env_params.harmonize_env_param_time(one, q).shape == (2,)


def test_env_params_ek_precedence_and_errors():
    d = synth.ek60_numpy(2, 6, 16)
    e = ed_mod.from_ek60_arrays(d)
    beam, env = e["Sonar/Beam_group1"], e["Environment"]
    out = env_params.get_env_params_EK("EK60", beam, env, {})
    np.testing.assert_array_equal(out["sound_speed"].values, d["sound_speed_indicative"])
    assert "temperature" not in out and "formula_absorption" not in out
    user = {"temperature": 8.0, "salinity": 34.0, "pressure": 50.0, "pH": 8.05}
    out = env_params.get_env_params_EK("EK60", beam, env, user)
    assert out["formula_sound_speed"] == "Mackenzie" and out["formula_absorption"] == "FG"
    assert out["sound_speed"] == pytest.approx(uwa.calc_sound_speed(8.0, 34.0, 50.0))
    out = env_params.get_env_params_EK("EK60", beam, env, {"sound_absorption": [0.01, 0.04]})
    np.testing.assert_array_equal(out["sound_absorption"].values, [0.01, 0.04])
    with pytest.raises(ValueError, match="sound_absorption"):
        env_params.get_env_params_EK("EK60", beam, env, {"sound_absorption": 0.01})
    with pytest.raises(ValueError, match="formula_absorption"):
        env_params.get_env_params_EK("EK60", beam, env, {"formula_absorption": "AZFP"})
    with pytest.raises(ValueError, match="'freq' is required"):
        env_params.get_env_params_EK("EK80", beam, env, {})
    with pytest.raises(ValueError, match="lengths of param value and channel do not match"):
        env_params.get_env_params_EK("EK60", beam, env, {"sound_absorption": [0.01]})


def test_cal_params_ek80_bb_scaling():
    d = synth.ek80_numpy(2, 3, 32)
    e = ed_mod.from_ek80_arrays(d, synth.ek80_filters())
    beam, vend = e["Sonar/Beam_group1"], e["Vendor_specific"]
    fc = DataArray(np.tile(((d["f_start"] + d["f_stop"]) / 2)[:, None], (1, 3)), ("channel", "ping_time"))
    out = cal_params.get_cal_params_EK("BB", fc, beam, vend, {}, sonar_type="EK80")
    fn = d["frequency_nominal"][:, None]
    np.testing.assert_allclose(out["equivalent_beam_angle"].values, d["psi"][:, None] + 20 * np.log10(fn / fc.values))
    np.testing.assert_allclose(out["beamwidth_alongship"].values, d["beamwidth_alongship"][:, None] * fn / fc.values)
    np.testing.assert_allclose(out["angle_offset_alongship"].values, np.tile(d["angle_offset_alongship"][:, None], (1, 3)))
    np.testing.assert_array_equal(out["impedance_transducer"].values, np.full((2, 3), 75.0))
    np.testing.assert_array_equal(np.asarray(out["impedance_transceiver"].values), d["z_er"])
    np.testing.assert_array_equal(out["receiver_sampling_frequency"].values, d["fs"])
    # frequency-dependent gain table from the user: interpolated at the centre frequency
    gtab = DataArray(np.array([[25.0, 27.0, 29.0]]), ("cal_channel_id", "cal_frequency"),
                     {"cal_channel_id": np.array([beam["channel"].values[0]]), "cal_frequency": np.array([45e3, 67.5e3, 90e3])})
    out = cal_params.get_cal_params_EK("BB", fc, beam, vend, {"gain_correction": gtab}, sonar_type="EK80")
    assert out["gain_correction"].values[0, 0] == pytest.approx(27.0)
    assert np.isnan(out["gain_correction"].values[1]).all()  # channel without a table, alternative = NaN
    with pytest.raises(ValueError, match="waveform_mode must be 'CW' or 'BB'"):
        cal_params.get_cal_params_EK("FM", fc, beam, vend, {})
    with pytest.raises(TypeError):
        cal_params.get_cal_params_EK(1, fc, beam, vend, {})


def test_azfp_rows_reproduce_oracle_range():
    d = synth.azfp_numpy(3, 5, 40)
    e = ed_mod.from_azfp_arrays(d)
    cal = CalibrateAZFP(e, {"salinity": d["salinity"], "pressure": d["pressure"]}, None)
    for ct in ("Sv", "TS"):
        rows = cal._rows(ct)
        s = np.arange(40)[None, None, :]
        R = (s * rows[..., 0:1]) * rows[..., 1:2] + rows[..., 2:3]
        ss = uwa.calc_sound_speed(d["temperature"], d["salinity"], d["pressure"], "AZFP")
        exp = ocal.range_azfp(40, cal_type=ct, sound_speed=np.tile(ss, (3, 1)), tau=d["transmit_duration_nominal"],
                              n_avg=d["number_of_samples_per_average_bin"], dig_rate=d["digitization_rate"],
                              lockout=d["lock_out_index"], C=3, P=5)
        np.testing.assert_allclose(R, exp, rtol=1e-13)
        np.testing.assert_allclose(rows[..., 7], -rows[..., 2] / rows[..., 1])


@pytest.mark.parametrize("bin_str", ["20s", "7s", "1min", "0.5h", "250ms", "2D", "90s", "1h"])
@pytest.mark.parametrize("start", ["2018-07-01T13:47:07.300000000", "2026-05-01T00:00:00", "2020-02-29T23:59:58.5"])
def test_resample_edges_match_pandas(bin_str, start):
    t0 = np.datetime64(start, "ns")
    pt = t0 + (np.arange(300) * 0.7e9).astype("timedelta64[ns]")
    e0, dt, n = gu.resample_edges(pt, bin_str)
    idx = pd.Series(0, index=pd.DatetimeIndex(pt)).resample(bin_str).first().index
    assert idx[0].value == e0 and len(idx) == n and pd.Timedelta(bin_str).value == dt


def test_bin_string_parsing_and_errors():
    assert gu._parse_x_bin("10m") == 10.0 and gu._parse_x_bin(" 0.5 M ") == 0.5
    with pytest.raises(TypeError, match="'x_bin' must be a string"):
        gu._parse_x_bin(10)
    with pytest.raises(ValueError, match="Range bin must be in meters"):
        gu._parse_x_bin("10km")
    with pytest.raises(KeyError):
        gu._parse_x_bin("10m", "foo")
    assert gu.ping_time_bin_parsing_and_conversion("20s") == (20, "second")
    assert gu.ping_time_bin_parsing_and_conversion("2min") == (2, "minute")
    assert extract_dB("3.0dB") == 3.0 and extract_dB("-120db") == -120.0
    with pytest.raises(TypeError):
        extract_dB(3.0)
    with pytest.raises(ValueError):
        extract_dB("3 dB")


def test_lite_containers():
    ds = Dataset(coords={"channel": ["a", "b"], "ping_time": np.arange(3)})
    ds["x"] = (("channel", "ping_time"), np.arange(6.0).reshape(2, 3), {"units": "m"})
    assert ds["x"].dims == ("channel", "ping_time") and ds["x"].attrs["units"] == "m"
    assert dict(ds.sizes) == {"channel": 2, "ping_time": 3} and "x" in ds and "ping_time" in ds
    np.testing.assert_array_equal(ds.isel(ping_time=slice(0, 2))["x"].values, [[0, 1], [3, 4]])
    with pytest.raises(ValueError, match="dimension"):
        ds["bad"] = (("channel",), np.arange(3.0))
    with pytest.raises(KeyError):
        ds["nope"]
    ds2 = ds.assign_attrs(a=1)
    assert ds2.attrs == {"a": 1} and ds.attrs == {}
    e = ed_mod.EchoData("EK60", {"Sonar/Beam_group1": ds})
    assert e["Sonar/Beam_group1"] is ds and "Platform" in e
    with pytest.raises(KeyError, match="no group"):
        e["Vendor_specific"]


# ---- noise masks / apply_mask: argument validation happens before any device work ----------------
def _mask_ds():
    ds = Dataset(coords={"channel": ["a", "b"], "ping_time": np.arange(4).astype("datetime64[s]"),
                         "range_sample": np.arange(5)})
    dims = ("channel", "ping_time", "range_sample")
    ds["Sv"] = (dims, np.zeros((2, 4, 5)))
    ds["echo_range"] = (dims, np.tile(np.arange(5.0), (2, 4, 1)))
    return ds


def test_mask_functions_validate_before_touching_the_gpu():
    from echopype_amd import clean

    ds = _mask_ds()
    for fn in (clean.mask_transient_noise, clean.mask_impulse_noise, clean.mask_attenuated_signal):
        with pytest.raises(ValueError, match="`range_var` must be either `echo_range` or `depth`."):
            fn(ds, range_var="range")
        with pytest.raises(ValueError, match="requires `depth` data variable in `ds_Sv`"):
            fn(ds)  # default range_var="depth" is absent
    with pytest.raises(ValueError, match="Input `func` is `nanmode`"):
        clean.mask_transient_noise(ds, func="nanmode", range_var="echo_range")
    with pytest.raises(TypeError, match="Decibal input must be a string"):
        clean.mask_impulse_noise(ds, impulse_noise_threshold=10.0, range_var="echo_range")
    with pytest.raises(ValueError, match="Range bin must be in meters"):
        clean.mask_impulse_noise(ds, depth_bin="5", range_var="echo_range")
    with pytest.raises(ValueError, match="Minimum range has to be shorter than maximum range"):
        clean.mask_attenuated_signal(ds, upper_limit_sl="180.0m", lower_limit_sl="170.0m", range_var="echo_range")


def test_apply_mask_validation_helpers():
    """mask/api.py:39-247 helpers, same error types and messages (tests/mask/test_mask.py)."""
    from echopype_amd.mask import api as mapi

    ds = _mask_ds()
    ok = DataArray(np.ones((4, 5), bool), ("ping_time", "range_sample"))
    assert mapi._validate_and_collect_mask_input(ok, {}) is ok
    assert len(mapi._validate_and_collect_mask_input([ok, ok], {})) == 2
    with pytest.raises(ValueError, match="single dict because mask is a single value"):
        mapi._validate_and_collect_mask_input(ok, [{}])
    with pytest.raises(TypeError, match="must be a list of dict or a dict"):
        mapi._validate_and_collect_mask_input([ok], 3)
    with pytest.raises(TypeError, match="must be a list of dict or a dict"):
        mapi._validate_and_collect_mask_input([ok], [{}, 3])
    with pytest.raises(ValueError, match="Masks must have one of the following dimensions"):
        mapi._validate_and_collect_mask_input(DataArray(np.ones((4, 5), bool), ("ping_time", "beam")), {})
    with pytest.raises(TypeError, match="Mask cannot contain NaN"):
        mapi._validate_and_collect_mask_input(DataArray(np.full((4, 5), np.nan), ok.dims), {})
    with pytest.raises(TypeError, match="Mask must be boolean"):
        mapi._validate_and_collect_mask_input(DataArray(np.full((4, 5), 3), ok.dims), {})
    with pytest.raises(NotImplementedError):
        mapi._validate_and_collect_mask_input("mask.zarr", {})
    with pytest.raises(ValueError, match="'channel' is a dimension in mask but not a dimension in source"):
        flat = Dataset(coords={"ping_time": ds["ping_time"].values, "range_sample": np.arange(5)})
        flat["Sv"] = (("ping_time", "range_sample"), np.zeros((4, 5)))
        mapi._check_mask_dim_alignment(flat, DataArray(np.ones((2, 4, 5), bool), ds["Sv"].dims), "Sv")
    with pytest.raises(ValueError, match="do not match the dimensions of source"):
        mapi._check_mask_dim_alignment(ds, DataArray(np.ones((4, 5), bool), ("ping_time", "depth")), "Sv")
    with pytest.raises(TypeError, match="var_name must be a string"):
        mapi._check_var_name_fill_value(ds, 1, np.nan)
    with pytest.raises(ValueError, match="does not contain the variable var_name"):
        mapi._check_var_name_fill_value(ds, "TS", np.nan)
    with pytest.raises(TypeError, match="fill_value must be of type int, float, or xr.DataArray"):
        mapi._check_var_name_fill_value(ds, "Sv", "x")
    with pytest.raises(ValueError, match="If fill_value is an array it must be of the same shape as Sv"):
        mapi._check_var_name_fill_value(ds, "Sv", DataArray(np.zeros((4, 4)), ok.dims))
    assert mapi._check_var_name_fill_value(ds, "Sv", 3) == 3


def test_geodesic_and_cumulative_distance_match_oracle():
    from oracle import nasc as onasc

    rng = np.random.default_rng(0)
    lat1, lat2 = rng.uniform(-80, 80, 50), rng.uniform(-80, 80, 50)
    lon1, lon2 = rng.uniform(-180, 180, 50), rng.uniform(-180, 180, 50)
    lat2[:25], lon2[:25] = lat1[:25] + rng.normal(0, 1e-3, 25), lon1[:25] + rng.normal(0, 1e-3, 25)  # ping spacing
    lat2[0], lon2[0] = lat1[0], lon1[0]
    got = gu.geodesic_distance_m(lat1, lon1, lat2, lon2)
    exp = np.array([onasc.geodesic_m(*p) for p in zip(lat1, lon1, lat2, lon2)])
    np.testing.assert_allclose(got, exp, rtol=1e-12, atol=1e-6)
    assert got[0] == 0.0
    lat = np.array([np.nan, 10.0, 10.001, np.nan, 10.003, 10.004, 10.004])
    lon = np.full(7, 20.0)
    ds = Dataset(coords={"ping_time": np.arange(7).astype("datetime64[s]")})
    ds["latitude"], ds["longitude"] = (("ping_time",), lat), (("ping_time",), lon)
    np.testing.assert_allclose(gu.get_distance_from_latlon(ds), onasc.distance_from_latlon(lat, lon), rtol=1e-12)
    ds["latitude"] = (("ping_time",), np.full(7, np.nan))
    with pytest.raises(ValueError, match="All lat/lon entries are NaN!"):
        gu.get_distance_from_latlon(ds)


# ---- add_depth EchoData-driven inputs: the reference's unit tests restated
#      (tests/consolidate/test_add_depth.py:51-187) --------------------------------------------------
def _hours(n, step):
    return np.datetime64("2024-07-04", "ns") + np.arange(n) * np.timedelta64(step, "h")


def test_ek_use_platform_vertical_offsets_output():
    from echopype_amd.consolidate import ek_depth_utils as eku

    plat = Dataset(coords={"time2": _hours(4, 5)})
    plat["water_level"] = (("time2",), np.array([1.5, 0.5, 0.0, 1.0]))
    plat["vertical_offset"] = (("time2",), np.array([1.0, 0.0, 0.0, 1.0]))
    plat["transducer_offset_z"] = (("time2",), np.array([3.0, 1.5, 0.0, 11.15]))
    depth, dims = eku.ek_use_platform_vertical_offsets(plat, _hours(5, 4))
    assert dims == ("ping_time",)
    np.testing.assert_allclose(depth, [0.5, 1.0, 0.0, 0.0, 9.15])
    # per-channel transducer offsets, scalar-in-time water level (EK80 layout)
    plat = Dataset(coords={"time2": _hours(4, 5), "channel": ["a", "b"]})
    plat["water_level"] = (("time2",), np.array([1.5, 0.5, 0.0, 1.0]))
    plat["vertical_offset"] = (("time2",), np.zeros(4))
    plat["transducer_offset_z"] = (("channel",), np.array([3.0, 5.0]))
    depth, dims = eku.ek_use_platform_vertical_offsets(plat, _hours(5, 4))
    assert dims == ("channel", "ping_time")
    np.testing.assert_allclose(depth, [[1.5, 2.5, 3.0, 3.0, 2.0], [3.5, 4.5, 5.0, 5.0, 4.0]])


def test_ek_use_platform_angles_output():
    from echopype_amd.consolidate import ek_depth_utils as eku

    plat = Dataset(coords={"time2": _hours(4, 5)})
    plat["pitch"] = (("time2",), np.array([-90, 0, 0, -45]))
    plat["roll"] = (("time2",), np.array([0, 90, 0, 0]))
    scaling, dims = eku.ek_use_platform_angles(plat, _hours(5, 4))
    assert dims == ("ping_time",)
    np.testing.assert_allclose(scaling, [0.0, 0.0, 1.0, 1.0, 1 / np.sqrt(2)], atol=1e-15)
    from scipy.spatial.transform import Rotation as R  # what the reference evaluates (ek_depth_utils.py:68-72)

    rng = np.random.default_rng(0)
    p, r = rng.uniform(-60, 60, 20), rng.uniform(-60, 60, 20)
    plat = Dataset(coords={"time2": _hours(20, 1)})
    plat["pitch"], plat["roll"] = (("time2",), p), (("time2",), r)
    scaling, _ = eku.ek_use_platform_angles(plat, _hours(20, 1))
    exp = R.from_euler("ZYX", np.column_stack([np.zeros(20), p, r]), degrees=True).as_matrix()[:, -1, -1]
    np.testing.assert_allclose(scaling, exp, rtol=1e-13)


def test_ek_use_beam_angles_output_and_warnings(caplog):
    import logging

    from echopype_amd.consolidate import ek_depth_utils as eku

    beam = Dataset(coords={"channel": ["chan1", "chan2", "chan3", "chan4"]})
    beam["beam_direction_x"] = (("channel",), np.array([1, 0, 0, 1]))
    beam["beam_direction_y"] = (("channel",), np.array([0, 1, 0, 0]))
    beam["beam_direction_z"] = (("channel",), np.array([0, 0, 1, np.sqrt(3) / 2]))
    with caplog.at_level(logging.WARNING):
        scaling, dims = eku.ek_use_beam_angles(beam)
    assert "Beam direction vector was not normalized" in caplog.text and dims == ("channel",)
    np.testing.assert_allclose(scaling, [0.0, 0.0, 1.0, (np.sqrt(3) / 2) / np.sqrt((np.sqrt(3) / 2) ** 2 + 1)])
    caplog.clear()
    beam = Dataset(coords={"channel": ["a", "b"]})
    beam["beam_direction_x"] = (("channel",), np.array([0, 0]))
    beam["beam_direction_y"] = (("channel",), np.array([0, 1]))
    beam["beam_direction_z"] = (("channel",), np.array([0, 0]))
    with caplog.at_level(logging.WARNING):
        scaling, _ = eku.ek_use_beam_angles(beam)
    assert "Some beam direction vectors are zero" in caplog.text
    assert np.isnan(scaling[0]) and scaling[1] == 0.0
    caplog.clear()
    beam["beam_direction_z"] = (("channel",), np.array([np.nan, 1.0]))
    with caplog.at_level(logging.WARNING):
        eku.ek_use_beam_angles(beam)
    assert "`beam_direction_z` variable array contains NaNs" in caplog.text


def test_bin_string_parsers_match_reference_outputs():
    """_parse_x_bin / ping_time_bin_parsing_and_conversion against the table the reference's own functions
    produced (oracle/gen_mvbs_index_goldens.py): accepted values, exception types and x-bin messages.
    Strings pandas turns into a zero-length Timedelta ('20', the integer 20) are rejected here (the reference
    fails on them one line later, in resample)."""
    import ast
    import os

    from echopype_amd.commongrid import utils as u

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_mvbs_index_goldens.npz"))
    n = 0
    for label, v, kind, res in g["parse_rows"].tolist():
        val = ast.literal_eval(v)
        if label == "ping_time_bin":
            if kind != "ok" or res.startswith("(0,"):
                with pytest.raises((ValueError, TypeError)):
                    u.ping_time_bin_parsing_and_conversion(val)
            else:
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    assert repr(u.ping_time_bin_parsing_and_conversion(val)) == res, v
        elif kind == "ok":
            assert repr(float(u._parse_x_bin(val, label))) == res, v
        else:
            with pytest.raises({"ValueError": ValueError, "TypeError": TypeError}[kind]) as e:
                u._parse_x_bin(val, label)
            assert str(e.value) == res, v
        n += 1
    assert n >= 30


def test_coarsen_time_mean_matches_reference_labels():
    import os

    from echopype_amd.commongrid.utils import coarsen_time_mean

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_mvbs_index_goldens.npz"))
    for tag in ("ix0", "ix1", "ix2", "ix3"):
        np.testing.assert_array_equal(coarsen_time_mean(g[f"{tag}_ping_time"], int(g[f"{tag}_args"][1])),
                                      g[f"{tag}_out_ping_time"])
    t = g["ix0_ping_time"].copy()
    t[:3] = np.datetime64("NaT")  # an all-NaT window and a partly NaT one
    t[4] = np.datetime64("NaT")
    out = coarsen_time_mean(t, 3)
    assert np.isnat(out[0]) and out[1] == t[3] + (t[5] - t[3]) // 2


def test_ek_depth_utils_match_the_reference_functions():
    """tests/golden/ref_depth_goldens.npz: outputs of the reference's own consolidate/ek_depth_utils.py
    (oracle/gen_depth_goldens.py) -- vertical offsets and pitch / roll scaling on an identical time axis and on a
    single time, beam-angle scaling incl. zero / tiny / NaN / unnormalised vectors."""
    import os

    from echopype_amd.consolidate import ek_depth_utils as eku

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_depth_goldens.npz"))
    pt = np.datetime64("2026-05-01T00:00:00", "ns") + np.arange(9) * np.timedelta64(3, "s")
    for tag in ("same", "single"):
        t2 = g[f"vo_{tag}_time2"]
        plat = Dataset(coords={"time2": t2})
        for k in ("water_level", "vertical_offset", "transducer_offset_z"):
            plat[k] = (("time2",), g[f"vo_{tag}_{k}"])
        depth, dims = eku.ek_use_platform_vertical_offsets(plat, pt)
        assert dims == ("ping_time",)
        np.testing.assert_array_equal(depth, g[f"vo_{tag}_out"])
        plat = Dataset(coords={"time2": g[f"pa_{tag}_time2"]})
        plat["pitch"], plat["roll"] = (("time2",), g[f"pa_{tag}_pitch"]), (("time2",), g[f"pa_{tag}_roll"])
        scaling, dims = eku.ek_use_platform_angles(plat, pt)
        assert dims == ("ping_time",)
        np.testing.assert_allclose(scaling, g[f"pa_{tag}_out"], rtol=0, atol=5e-16)  # cos*cos vs a rotation matrix: 2 ulp
        np.testing.assert_array_equal(np.isnan(scaling), np.isnan(g[f"pa_{tag}_out"]))
    v = g["ba_vectors"]
    beam = Dataset(coords={"channel": [f"ch{i}" for i in range(len(v))]})
    for i, k in enumerate(("beam_direction_x", "beam_direction_y", "beam_direction_z")):
        beam[k] = (("channel",), v[:, i])
    scaling, dims = eku.ek_use_beam_angles(beam)
    assert dims == ("channel",)
    np.testing.assert_array_equal(scaling, g["ba_out"])


# ---- a26: the > 2 GiB backscatter warning and the env_params / cal_params type check ---------------------------
def test_check_echodata_backscatter_size_warning(caplog):
    """tests/calibrate/test_calibrate.py:342-441 of the reference: the calibrator warns, with exactly this text, when
    the backscatter variables exceed 2 GiB (calibrate_base.py:95-128).  The large array is a zero-stride broadcast
    view: it reports 2.2 GiB without occupying them."""
    import logging

    import echopype_amd as ep
    from echopype_amd.calibrate.api import CALIBRATOR

    expected = (
        "The Echodata backscatter variables are large and can cause memory issues. "
        "Consider modifying the workflow that uses compute_Sv as below: "
        "Prior to `compute_Sv` run `echodata.chunk(CHUNK_DICTIONARY) "
        "and after `compute_Sv` run `ds_Sv.to_zarr(ZARR_STORE, compute=True)`. "
        "This will ensure that the computation is lazily evaluated, "
        "with the results stored directly in a Zarr store on disk, rather then in memory.")
    for P, warns in ((300_000, True), (200, False)):
        d = ep.synth.ek60_numpy(2, 4, 8)
        for k, v in list(d.items()):
            if isinstance(v, np.ndarray) and v.ndim == 2 and v.shape == (2, 4):
                d[k] = np.repeat(v[:, :1], P, axis=1)
        d["ping_time"] = ep.synth.T0 + (np.arange(P) * 1_000_000_000).astype("timedelta64[ns]")
        d["backscatter_r"] = np.broadcast_to(np.float32(-70.0), (2, P, 1000))
        assert (d["backscatter_r"].nbytes / 1024**3 > 2.0) == warns
        cal = CALIBRATOR["EK60"](ep.echodata.from_ek60_arrays(d), env_params=None, cal_params=None, ecs_file=None)
        caplog.clear()
        with caplog.at_level(logging.WARNING, logger="echopype_amd.calibrate"):
            cal._check_echodata_backscatter_size()
        msgs = [r.message for r in caplog.records]
        assert msgs == ([expected] if warns else [])


@pytest.mark.parametrize("bad", ["env_params", "cal_params"])
def test_env_and_cal_params_must_be_dicts(bad):
    """calibrate_base.py:35-47: anything but None or a dict is rejected with the reference's message."""
    import echopype_amd as ep
    from echopype_amd.calibrate.api import CALIBRATOR

    ed = ep.echodata.from_ek60_arrays(ep.synth.ek60_numpy(2, 5, 16))
    kw = dict(env_params=None, cal_params=None, ecs_file=None)
    kw[bad] = [("sound_speed", 1500.0)]
    with pytest.raises(ValueError, match=f"'{bad}' has to be None or a dict"):
        CALIBRATOR["EK60"](ed, **kw)


def test_env_params_ek80_formulas_on_time1_then_onto_ping_time():
    """Several Environment timestamps (a merged EK80 dataset): the reference evaluates absorption on ``time1`` and
    harmonises the RESULT (env_params.py:300-351), i.e. out = interp_time1->ping_time( f(T(time1), ...) ), which is
    not f(interp(T), ...).  Checked against the formula itself (pinned by the reference leaf goldens) evaluated at the
    Environment timestamps and interpolated linearly, extrapolation included."""
    d = synth.ek80_numpy(2, 7, 32)
    e = ed_mod.from_ek80_arrays(d, synth.ek80_filters())
    beam, env0 = e["Sonar/Beam_group1"], e["Environment"]
    t0 = np.asarray(beam["ping_time"].values)[0]
    time1 = t0 + np.array([-2, 1, 9]) * np.timedelta64(1, "s")  # the first ping lies before, the last after... inside
    T1, S1, D1, pH1, ss1 = (np.array(v, float) for v in ([4.0, 12.0, 18.0], [33.0, 34.5, 35.0], [5.0, 60.0, 200.0],
                                                          [7.9, 8.0, 8.1], [1470.0, 1495.0, 1510.0]))
    chans = np.asarray(beam["channel"].values)
    env = ed_mod.Dataset(coords={"time1": time1, "channel": chans})
    for k, v in (("temperature", T1), ("salinity", S1), ("depth", D1), ("acidity", pH1), ("sound_speed_indicative", ss1)):
        env[k] = (("time1",), v)
    freq = DataArray(d["frequency_nominal"], ("channel",))
    out = env_params.get_env_params_EK("EK80", beam, env, {}, freq=freq)
    x = time1.astype("datetime64[ns]").astype(np.int64).astype(float)
    xq = np.asarray(beam["ping_time"].values).astype("datetime64[ns]").astype(np.int64).astype(float)
    ab1 = uwa.calc_absorption(frequency=d["frequency_nominal"][:, None], temperature=T1, salinity=S1, pressure=D1, pH=pH1,
                              sound_speed=ss1, formula_source="FG")  # (channel, time1)
    hi = np.clip(np.searchsorted(x, xq, side="left"), 1, 2)
    lo = hi - 1
    exp = ab1[:, lo] + (ab1[:, hi] - ab1[:, lo]) * (xq - x[lo]) / (x[hi] - x[lo])
    np.testing.assert_allclose(out["sound_absorption"].values, exp, rtol=1e-13)
    naive = uwa.calc_absorption(frequency=d["frequency_nominal"][:, None], temperature=np.interp(xq, x, T1),
                                salinity=np.interp(xq, x, S1), pressure=np.interp(xq, x, D1), pH=np.interp(xq, x, pH1),
                                sound_speed=np.interp(xq, x, ss1), formula_source="FG")
    assert np.abs(naive - exp).max() > 1e-6 * np.abs(exp).max()  # the two orders of operation really differ here
    # user T/S/P/pH on time1: sound speed by formula on time1, then onto ping_time
    user = {k: DataArray(v, ("time1",), {"time1": time1, "channel": chans})
            for k, v in (("temperature", T1), ("salinity", S1), ("pressure", D1), ("pH", pH1))}
    ssf = uwa.calc_sound_speed(T1, S1, D1)
    exp_ss = ssf[lo] + (ssf[hi] - ssf[lo]) * (xq - x[lo]) / (x[hi] - x[lo])
    try:
        out2 = env_params.get_env_params_EK("EK80", beam, env, user, freq=freq)
    except ValueError:
        return  # user DataArrays must carry every channel as a coordinate: covered by the sanitiser's own tests
    np.testing.assert_allclose(out2["sound_speed"].values, exp_ss, rtol=1e-13)


def test_env_params_multi_timestamp_environment_with_a_per_channel_user_parameter():
    """Several Environment timestamps AND a per-channel user parameter (a pH per channel): the per-channel value
    cannot ride the time1 path -- the parameters are harmonised onto ping_time and the (channel,) value broadcasts as
    (C, 1) against the (P,) ones (the advisor's round-2 finding: it came back as (C,) and mis-broadcast)."""
    d = synth.ek80_numpy(2, 7, 32)
    e = ed_mod.from_ek80_arrays(d, synth.ek80_filters())
    beam = e["Sonar/Beam_group1"]
    t0 = np.asarray(beam["ping_time"].values)[0]
    time1 = t0 + np.array([-2, 1, 9]) * np.timedelta64(1, "s")
    T1, S1, D1, pH1, ss1 = (np.array(v, float) for v in ([4.0, 12.0, 18.0], [33.0, 34.5, 35.0], [5.0, 60.0, 200.0],
                                                          [7.9, 8.0, 8.1], [1470.0, 1495.0, 1510.0]))
    chans = np.asarray(beam["channel"].values)
    env = ed_mod.Dataset(coords={"time1": time1, "channel": chans})
    for k, v in (("temperature", T1), ("salinity", S1), ("depth", D1), ("acidity", pH1), ("sound_speed_indicative", ss1)):
        env[k] = (("time1",), v)
    freq = DataArray(d["frequency_nominal"], ("channel",))
    pH_c = np.array([7.8, 8.2])
    out = env_params.get_env_params_EK("EK80", beam, env, {"pH": DataArray(pH_c, ("channel",), {"channel": chans})},
                                       freq=freq)
    x = time1.astype("datetime64[ns]").astype(np.int64).astype(float)
    xq = np.asarray(beam["ping_time"].values).astype("datetime64[ns]").astype(np.int64).astype(float)

    def lin(v):  # linear inter- / extrapolation onto ping_time, as harmonize_env_param_time does
        hi = np.clip(np.searchsorted(x, xq, side="left"), 1, 2)
        lo = hi - 1
        return v[lo] + (v[hi] - v[lo]) * (xq - x[lo]) / (x[hi] - x[lo])

    exp = uwa.calc_absorption(frequency=d["frequency_nominal"][:, None], temperature=lin(T1)[None, :],
                              salinity=lin(S1)[None, :], pressure=lin(D1)[None, :], pH=pH_c[:, None],
                              sound_speed=lin(ss1)[None, :], formula_source="FG")
    assert out["sound_absorption"].shape == (2, 7)
    np.testing.assert_allclose(out["sound_absorption"].values, exp, rtol=1e-12)
    assert np.abs(exp[0] - exp[1]).max() > 0  # (the channels really differ)


def test_lazy_device_array_bookkeeping():
    """xr_lite.LazyDeviceArray (the echo_range compute_Sv leaves behind) on CPU tensors: shape / dtype / size and the
    statistics are known without producing the array; the producer runs once, on first read; the coefficient rows,
    the NaN source and the statistics stop being offered once the array (or the source) has been written to."""
    import torch

    from echopype_amd.xr_lite import DataArray, DeviceArray, LazyDeviceArray

    calls = []
    raw = torch.tensor([[[1.0, float("nan"), 3.0, 4.0]]])
    rows = torch.zeros((1, 1, 8), dtype=torch.float64)

    def make():
        calls.append(1)
        out = torch.arange(4, dtype=torch.float64).reshape(1, 1, 4) * 0.5
        return torch.where(torch.isnan(raw), torch.full_like(out, float("nan")), out)

    stats = torch.tensor([0.0, 1.5, 1.0], dtype=torch.float64)
    lz = LazyDeviceArray((1, 1, 4), torch.float64, raw.device, make, stats=stats, rows=rows, nan_where=raw)
    assert isinstance(lz, DeviceArray) and not lz.materialized
    assert lz.shape == (1, 1, 4) and lz.ndim == 3 and lz.dtype == np.dtype("float64") and lz.nbytes == 32
    assert lz.cached_stats() == (0.0, 1.5, 1) and lz.coef_rows() is rows and lz.nan_source() is raw
    da = DataArray(lz, ("channel", "ping_time", "range_sample"))
    assert da.shape == (1, 1, 4) and not lz.materialized and calls == []
    np.testing.assert_array_equal(da.values, [[[0.0, np.nan, 1.0, 1.5]]])
    assert lz.materialized and calls == [1]
    da.values
    assert calls == [1]                                           # produced once
    assert lz.cached_stats() == (0.0, 1.5, 1) and lz.coef_rows() is rows
    lz.tensor[0, 0, 0] = 7.0                                      # written to: no longer the function of its rows
    assert lz.cached_stats() is None and lz.coef_rows() is None
    lz3 = LazyDeviceArray((1, 1, 4), torch.float64, raw.device, make, rows=rows, nan_where=raw)  # no statistics (fused path)
    assert lz3.coef_rows() is rows and lz3.cached_stats() is None
    lz3.tensor.add_(1.0)
    assert lz3.coef_rows() is None                                # ... the rows are withdrawn all the same
    lz2 = LazyDeviceArray((1, 1, 4), torch.float64, raw.device, make, stats=stats, rows=rows, nan_where=raw)
    raw[0, 0, 0] = float("nan")                                   # the NaN source was written to
    assert lz2.nan_source() is None and lz2.coef_rows() is rows
    assert "lazy" in repr(lz2) and "materialized" in repr(lz)


def test_lazy_device_array_fulfil_and_statistics_hook():
    """The deferred Sv of compute_Sv: a consumer that produced the array in its own pass installs it with fulfil() (once,
    shape and dtype checked; the producer never runs); statistics asked for before anybody left them trigger the hook once."""
    import torch

    from echopype_amd.xr_lite import LazyDeviceArray

    calls = []

    def make():
        calls.append("make")
        return torch.ones((2, 3), dtype=torch.float64)

    src = object()
    sv = LazyDeviceArray((2, 3), torch.float64, torch.device("cpu"), make, source=src)
    assert sv.source is src and not sv.materialized
    with pytest.raises(ValueError):
        sv.fulfil(torch.zeros((2, 4), dtype=torch.float64))
    with pytest.raises(ValueError):
        sv.fulfil(torch.zeros((2, 3), dtype=torch.float32))
    t = torch.full((2, 3), 5.0, dtype=torch.float64)
    sv.fulfil(t)
    assert sv.materialized and sv.tensor is t and sv.source is None and calls == []
    with pytest.raises(RuntimeError):
        sv.fulfil(t)
    # the range variable that travels with it: no statistics yet, a hook that produces them (here: reading the Sv)
    sv2 = LazyDeviceArray((2, 3), torch.float64, torch.device("cpu"), make, source=src)
    rng = LazyDeviceArray((2, 3), torch.float64, torch.device("cpu"), lambda: torch.zeros((2, 3), dtype=torch.float64))

    def hook():
        calls.append("hook")
        sv2.tensor
        rng.set_stats(torch.tensor([0.0, 2.0, 1.0], dtype=torch.float64))

    rng.set_stats(None, hook=hook)
    assert calls == []
    assert rng.cached_stats() == (0.0, 2.0, 1) and calls == ["hook", "make"] and sv2.materialized and not rng.materialized
    assert rng.cached_stats() == (0.0, 2.0, 1) and calls == ["hook", "make"]      # asked again: nothing runs
    rng.tensor.add_(1.0)                                                           # written to: the statistics are void
    assert rng.cached_stats() is None


def test_resample_edges_sorted_shortcut_equals_the_scan():
    """compute_MVBS on a deferred Sv has already checked its ping times (non-decreasing, no NaT) and takes the bin edges
    from the two ends: the same edges as the scan over all of them."""
    from echopype_amd.commongrid.utils import resample_edges

    rng = np.random.default_rng(3)
    for bin_ in ("1s", "20s", "7min", "1h"):
        for _ in range(20):
            n = int(rng.integers(1, 400))
            t0 = np.datetime64("2026-03-09T00:00:00", "ns") + np.timedelta64(int(rng.integers(0, 86400 * 3)), "s")
            t = t0 + np.sort(rng.integers(0, 10**12, n)).astype("timedelta64[ns]")
            assert resample_edges(t, bin_, sorted_valid=True) == resample_edges(t, bin_)


def test_deferred_dataset_assembles_once_on_first_use():
    """xr_lite.DeferredDataset: a Dataset whose build() runs when anything touches it -- once; it IS a Dataset for every
    isinstance check of the package, forwards reads and writes to the built object, and lets a failing build surface at
    that first use (and again, the same error, on later ones)."""
    from echopype_amd.xr_lite import DataArray, Dataset, DeferredDataset

    calls = []

    def build():
        calls.append(1)
        ds = Dataset(coords={"x": np.arange(3)}, attrs={"a": 1})
        ds["v"] = (("x",), np.array([1.0, 2.0, 3.0]))
        return ds

    dd = DeferredDataset(build)
    assert isinstance(dd, Dataset) and not dd.resolved and not calls
    assert "v" in dd and calls == [1] and dd.resolved
    assert dd.sizes["x"] == 3 and list(dd.data_vars) == ["v"] and dd.attrs == {"a": 1} and calls == [1]
    np.testing.assert_array_equal(dd["v"].values, [1.0, 2.0, 3.0])
    np.testing.assert_array_equal(dd.v.values, [1.0, 2.0, 3.0])            # attribute access like xarray's
    dd["w"] = DataArray(np.zeros(3), ("x",))                               # writes land in the built dataset
    dd.attrs["b"] = 2
    assert set(dd.data_vars) == {"v", "w"} and dd.attrs == {"a": 1, "b": 2}
    cp = dd.assign_attrs(c=3)
    assert type(cp) is Dataset and cp.attrs["c"] == 3 and "c" not in dd.attrs
    assert repr(dd).startswith("<Dataset") and calls == [1]

    def bad():
        raise ValueError("range bins are empty")

    bd = DeferredDataset(bad)
    with pytest.raises(ValueError, match="range bins are empty"):
        bd.sizes
    with pytest.raises(ValueError, match="range bins are empty"):  # the SAME error again, not a generic "failed earlier"
        bd["v"]


def test_lazy_attrs_settle_on_first_read_of_a_value_only():
    """xr_lite.LazyAttrs (the ``actual_range`` of remove_background_noise's deferred outputs): keys, len and ``in``
    never run the thunk; any read of the value does, once, whatever the route (``[]``, get, items, dict(), ``**``,
    ``==``, json, pickle, copies); DataArray / Dataset construction keeps the thunk; a write replaces it."""
    import copy
    import json
    import pickle

    from echopype_amd.xr_lite import DataArray, Dataset, LazyAttrs

    calls = []

    def make():
        a = LazyAttrs({"long_name": "x", "units": "dB"})
        a.set_lazy("actual_range", lambda: (calls.append(1), [1.0, 2.0])[1])
        a["n"] = 3
        return a

    a = make()
    assert list(a) == ["long_name", "units", "actual_range", "n"] and len(a) == 4 and "actual_range" in a
    assert list(a.keys()) == list(a) and a.has_pending("actual_range") and not calls
    ds = Dataset(coords={"x": np.arange(3)})
    ds["v"] = DataArray(np.zeros(3), ("x",), attrs=a)
    v, cp = ds["v"], ds.copy()["v"].copy()
    assert isinstance(v.attrs, LazyAttrs) and v.attrs.has_pending() and cp.attrs.has_pending() and not calls
    assert cp.attrs["actual_range"] == [1.0, 2.0] and calls == [1]
    assert a["actual_range"] == [1.0, 2.0] and v.attrs.get("actual_range") == [1.0, 2.0] and calls == [1]  # shared, once
    assert dict(make()) == {"long_name": "x", "units": "dB", "actual_range": [1.0, 2.0], "n": 3}
    assert {**make()}["actual_range"] == [1.0, 2.0] and make() == dict(a) and dict(a) == make()
    assert json.loads(json.dumps(make()))["actual_range"] == [1.0, 2.0]
    assert pickle.loads(pickle.dumps(make())) == dict(a) and type(copy.deepcopy(make())) is dict
    assert list(make().items())[2] == ("actual_range", [1.0, 2.0]) and [1.0, 2.0] in make().values()
    assert "1.0, 2.0" in repr(make()) and make().pop("actual_range") == [1.0, 2.0]
    n = len(calls)
    w = make()
    w["actual_range"] = [0, 0]
    u = make()
    u.update(actual_range=5)
    d = make()
    del d["actual_range"]
    assert w["actual_range"] == [0, 0] and u["actual_range"] == 5 and "actual_range" not in d and len(calls) == n
    assert not w.has_pending() and not u.has_pending() and not d.has_pending()


def test_deferred_dataset_keeps_errors_but_retries_after_an_interrupt():
    """xr_lite.DeferredDataset: an error of the assembly is THE result (raised again at every access); an interrupt is
    not -- the next access assembles again."""
    from echopype_amd.xr_lite import Dataset, DeferredDataset

    calls = []

    def build():
        calls.append(1)
        if len(calls) == 1:
            raise KeyboardInterrupt
        return Dataset(coords={"x": np.arange(3)})

    ds = DeferredDataset(build)
    with pytest.raises(KeyboardInterrupt):
        ds.coords
    assert list(ds.coords["x"].values) == [0, 1, 2] and len(calls) == 2

    def fails():
        raise ValueError("range bins are empty")

    bad = DeferredDataset(fails)
    for _ in range(2):
        with pytest.raises(ValueError, match="range bins are empty"):
            bad.attrs
