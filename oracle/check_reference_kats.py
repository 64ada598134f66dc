#!/usr/bin/env python3
"""The reference's OWN functions, executed from /root/reference over oracle/xr_shim.py, on the inputs the reference's
OWN tests build, held to the expectations written in those test files (authoring container only: needs
/root/reference; run by tests/test_reference_kats_over_shim.py when it is there).

The golden generators (oracle/gen_*goldens.py) run the same reference code over the same stand-in for xarray; this
script closes the loop "stand-in + reference code == numbers the reference's maintainers wrote down":

  clean/api.py::remove_background_noise       tests/clean/test_noise.py:902-987  (the two noise points are NaN; on the
                                              seed-1 normal data exactly 6 of the first 50 range samples are removed)
  commongrid/api.py::compute_MVBS_index_binning
                                              tests/commongrid/test_commongrid_api.py:171-202 (shape, and array_equal
                                              with 10 log10 of the padded block mean of 10^(Sv/10))
  calibrate/cal_params.py::get_vend_cal_params_power
                                              tests/calibrate/test_cal_params.py:751-868 (four tables: with / without
                                              NaN pulse lengths, channel order equal / different in Vendor_specific)
  calibrate/env_params.py::harmonize_env_param_time (+ utils/align.py::align_to_ping_time)
                                              tests/calibrate/test_env_params.py:29-126 (scalar, one timestamp, identical
                                              axis, the 0.5 and [0.5, 2880.5] interpolations)
Exit status 0 = every expectation met."""
import logging
import os
import sys
import types

import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import xr_shim  # noqa: E402
from gen_goldens import REF, _load  # noqa: E402
from gen_maskapi_goldens import load_clean_api  # noqa: E402  (stub modules + the reference's clean/api.py)

DA, DS = xr_shim.DataArray, xr_shim.Dataset
DIMS = ["channel", "ping_time", "range_sample"]


def check_remove_background_noise(api):
    nchan, npings, nrange = 1, 10, 100
    chan = np.arange(nchan).astype(str)
    ping_time = pd.date_range(start="2015-06-30T17:30:10.000000000", end="2015-06-30T17:30:25.000000000",
                              periods=npings).to_numpy()
    data = np.ones(nrange)
    np.put(data, 30, -30)
    np.put(data, 60, -30)
    data = np.array([data] * npings)

    def dataset(sv, rmax):
        ds = DS(coords={"channel": chan, "ping_time": ping_time, "range_sample": np.arange(nrange)})
        ds["Sv"] = DA(sv, dims=DIMS)
        ds["echo_range"] = DA(np.array([[np.linspace(0, rmax, nrange)] * npings]), dims=DIMS)
        ds["sound_absorption"] = 0.001
        return ds

    out = api.remove_background_noise(dataset(np.array([data]), 10), ping_num=2, range_sample_num=5, SNR_threshold="0dB")
    sc = out["Sv_corrected"].transpose(*DIMS).data
    assert np.isnan(sc[0, 0, 30]) and np.isnan(sc[0, 0, 60])
    np.random.seed(1)
    data = np.random.normal(loc=-100, scale=2, size=(nchan, npings, nrange))
    out = api.remove_background_noise(dataset(data, 3), ping_num=2, range_sample_num=5, SNR_threshold="0dB")
    null = np.isnan(out["Sv_corrected"].transpose(*DIMS).data)
    assert np.count_nonzero(null[0, :, :50]) == 6, np.count_nonzero(null[0, :, :50])
    return "remove_background_noise: noise points NaN; 6 of the first 50 range samples removed on the seed-1 data"


def check_index_binning():
    cg_api = _load("echopype.commongrid.api", f"{REF}/commongrid/api.py")
    # tests/conftest.py::regular_data_params / mock_data.py::_gen_Sv_echo_range_regular: 2 channels, 4 x ... the shape and
    # the random values do not matter to the assertion (it is an identity); use the fixture's sizes
    nchan, npings, nrange, ping_num, rsn = 2, 120, 13, 3, 7
    rng = np.random.default_rng(0)
    sv = rng.uniform(-100, -20, (nchan, npings, nrange))
    chans = np.array([f"ch_{i}" for i in range(nchan)])
    pings = np.datetime64("2020-01-01T01:00:00", "ns") + np.arange(npings) * np.timedelta64(1, "s")
    ds = DS(coords={"channel": chans, "ping_time": pings, "range_sample": np.arange(nrange)})
    ds["Sv"] = DA(sv, dims=DIMS)
    ds["echo_range"] = DA(np.tile(np.arange(nrange) * 0.5, (nchan, npings, 1)), dims=DIMS)
    ds["frequency_nominal"] = DA(np.array([38e3, 120e3]), {"channel": chans}, ["channel"])
    out = cg_api.compute_MVBS_index_binning(ds, range_sample_num=rsn, ping_num=ping_num)
    got = out["Sv"].transpose(*DIMS).data
    shape = np.ceil((nchan, npings / ping_num, nrange / rsn)).astype(int)
    assert np.all(got.shape == shape), (got.shape, shape)
    # the test's expectation, evaluated with NumPy alone: pad to whole blocks with NaN, nanmean per block
    lin = 10 ** (sv / 10)
    P2, S2 = shape[1] * ping_num, shape[2] * rsn
    pad = np.full((nchan, P2, S2), np.nan)
    pad[:, :npings, :nrange] = lin
    exp = 10 * np.log10(np.nanmean(pad.reshape(nchan, shape[1], ping_num, shape[2], rsn), axis=(2, 4)))
    assert np.array_equal(got, exp)
    return f"compute_MVBS_index_binning: shape {tuple(shape)}, array_equal with the padded block mean"


def check_vend_cal_params_power():
    for name, attrs in (("ecs", ["ECSParser", "conform_channel_order", "ecs_ds2dict", "ecs_ev2ep"]),):
        m = types.ModuleType(f"echopype.calibrate.{name}")
        for a in attrs:
            setattr(m, a, None)
        sys.modules.setdefault(m.__name__, m)
    for n, p in (("echopype.calibrate", [f"{REF}/calibrate"]), ("echopype.echodata", [])):
        if n not in sys.modules:
            m = types.ModuleType(n)
            m.__path__ = p
            sys.modules[n] = m
    sys.modules["echopype.echodata"].EchoData = object
    cp = _load("echopype.calibrate.cal_params_kat", f"{REF}/calibrate/cal_params.py")
    vend = DS(coords={"channel": np.array(["chA", "chB"]), "pulse_length_bin": np.arange(4)})
    for p in ("sa_correction", "gain_correction"):
        vend[p] = DA(np.array([[10, 20, 30, 40], [110, 120, 130, 140]]),
                     {"channel": np.array(["chA", "chB"]), "pulse_length_bin": np.arange(4)}, ["channel", "pulse_length_bin"])
    vend["pulse_length"] = DA(np.array([[64, 128, 256, 512], [128, 256, 512, 1024]]),
                              {"channel": np.array(["chA", "chB"]), "pulse_length_bin": np.arange(4)},
                              ["channel", "pulse_length_bin"])
    nan = np.nan
    cases = [
        ([[64, 256, 128, 512], [512, 1024, 256, 128]], ["chA", "chB"], [[10, 30, 20, 40], [130, 140, 120, 110]]),
        ([[512, 1024, 256, 128], [64, 256, 128, 512]], ["chB", "chA"], [[130, 140, 120, 110], [10, 30, 20, 40]]),
        ([[64, nan, 128, 512], [512, 1024, 256, nan]], ["chA", "chB"], [[10, nan, 20, 40], [130, 140, 120, nan]]),
        ([[512, 1024, 256, nan], [64, nan, 128, 512]], ["chB", "chA"], [[130, 140, 120, nan], [10, nan, 20, 40]]),
    ]
    for tdn, chans, exp in cases:
        beam = DS(coords={"ping_time": np.array([1, 2, 3, 4]), "channel": np.array(chans)})
        beam["transmit_duration_nominal"] = DA(np.array(tdn, dtype=float).T, {"ping_time": np.array([1, 2, 3, 4]),
                                                                           "channel": np.array(chans)}, ["ping_time", "channel"])
        out = cp.get_vend_cal_params_power(beam, vend, "sa_correction")
        got = out.transpose("ping_time", "channel")
        assert list(got.coords["channel"]) == chans, (list(got.coords["channel"]), chans)
        np.testing.assert_allclose(got.data.astype(float), np.array(exp, dtype=float).T, equal_nan=True)
        assert "pulse_length_bin" not in got.coords
    return "get_vend_cal_params_power: the four pulse-length tables (NaN pulse lengths, permuted channels)"


def check_harmonize_env_param_time():
    for n, p in (("echopype.utils", []),):
        if n not in sys.modules:
            m = types.ModuleType(n)
            m.__path__ = p
            sys.modules[n] = m
    _load("echopype.utils.align", f"{REF}/utils/align.py")
    uw = types.ModuleType("echopype.utils.uwa")
    uw.calc_absorption = uw.calc_sound_speed = None
    sys.modules["echopype.utils.uwa"] = uw
    sys.modules["echopype.utils"].uwa = uw
    env = _load("echopype.calibrate.env_params_kat", f"{REF}/calibrate/env_params.py")
    h = env.harmonize_env_param_time
    assert h(p=10.05) == 10.05
    one = DA(np.array([2]), {"time1": np.array(["2017-06-20T01:00:00"], dtype="datetime64[ns]")}, ["time1"])
    assert h(p=one) == 2
    t3 = np.arange("2017-06-20T01:00:00", "2017-06-20T01:01:30", np.timedelta64(30, "s"), dtype="datetime64[ns]")
    p = DA(np.array([0, 1, 2]), {"time1": t3}, ["time1"])
    try:
        h(p=p, ping_time=None)
        raise AssertionError("ping_time=None must raise")
    except ValueError:
        pass
    same = p["time1"].rename({"time1": "ping_time"})
    new = h(p=p, ping_time=same)
    assert (new["ping_time"] == same).all() and (new.data == p.data).all()
    q = np.array(["2017-06-20T01:00:15"], dtype="datetime64[ns]")
    new = h(p=p, ping_time=DA(q, {"ping_time": q}, ["ping_time"]))
    assert (np.asarray(new.coords["ping_time"]) == q).all() and new.data == 0.5
    time1 = np.arange("2017-06-20T01:00:00", "2017-06-22T01:00:31", np.timedelta64(30, "s"), dtype="datetime64[ns]")
    p = DA(np.arange(len(time1)), {"time1": time1}, ["time1"])
    q = np.array(["2017-06-20T01:00:15", "2017-06-21T01:00:15"], dtype="datetime64[ns]")
    new = h(p=p, ping_time=DA(q, {"ping_time": q}, ["ping_time"]))
    assert np.array_equal(np.asarray(new.coords["ping_time"]), q) and (new.data == [0.5, 2880.5]).all(), new.data
    return "harmonize_env_param_time: scalar, one timestamp, identical axis, 0.5 and [0.5, 2880.5]"


def main():
    logging.disable(logging.WARNING)
    api = load_clean_api()
    lines = [check_remove_background_noise(api), check_index_binning(), check_vend_cal_params_power(),
             check_harmonize_env_param_time()]
    for ln in lines:
        print("ok  ", ln)
    return 0


if __name__ == "__main__":
    sys.exit(main())
