"""Oracle vs the reference's own synthetic known-answer tests (restated).  CPU only."""
import numpy as np
import pandas as pd
import pytest

import kat_fixtures as kf
from oracle import calibrate, clean, commongrid


def test_noise_toy_and_seed1():
    Sv, er, a = kf.noise_toy()
    _, corr = clean.remove_background_noise(Sv, er, a, 2, 5, SNR_threshold="0dB")
    assert np.isnan(corr[0, 0, 30]) and np.isnan(corr[0, 0, 60])
    Sv, er, a = kf.noise_seed1()
    _, corr = clean.remove_background_noise(Sv, er, a, 2, 5, SNR_threshold="0dB")
    assert np.count_nonzero(np.isnan(corr[0, :, :50])) == 6  # test_noise.py:983-987


def test_noise_upsampling_pairs_equal():
    # test_noise.py:865-899 property: with ping_num=2 consecutive ping pairs share one noise value
    Sv, er, a = kf.noise_seed1()
    sn = clean.estimate_background_noise(Sv, er, a, 2, 5)
    tl = sn - (20 * np.log10(np.where(er >= 1, er, 1)) + 2 * a * er)
    np.testing.assert_allclose(tl[:, 0::2, :], tl[:, 1::2, :], rtol=0, atol=1e-12)


def test_extract_dB_errors():
    assert clean.extract_dB("3.0dB") == 3.0 and clean.extract_dB("-120db") == -120.0
    with pytest.raises(TypeError):
        clean.extract_dB(3.0)
    with pytest.raises(ValueError):
        clean.extract_dB("3.0 dB")


@pytest.mark.parametrize("tau,expected", kf.PULSE_CASES)
def test_pulse_length_lookup(tau, expected):
    got = calibrate.vend_cal_params_power(tau, kf.PULSE_TABLE["pulse_length"], kf.PULSE_TABLE["table"])
    np.testing.assert_array_equal(got, expected)


def test_harmonize_time_interp():
    # tests/calibrate/test_env_params.py:70-126
    t1 = np.array(["2017-06-20T01:00:00", "2017-06-20T01:00:30", "2017-06-20T01:01:00"], "datetime64[ns]")
    q = np.array(["2017-06-20T01:00:15"], "datetime64[ns]")
    assert calibrate.harmonize_time(np.array([0.0, 1, 2]), t1, q)[0] == 0.5
    t1 = np.arange("2017-06-20T01:00:00", "2017-06-22T01:00:31", np.timedelta64(30, "s"), dtype="datetime64[ns]")
    q = np.array(["2017-06-20T01:00:15", "2017-06-21T01:00:15"], "datetime64[ns]")
    np.testing.assert_array_equal(calibrate.harmonize_time(np.arange(len(t1), dtype=float), t1, q), [0.5, 2880.5])
    # identical axis -> passthrough; single timestamp -> squeeze
    np.testing.assert_array_equal(calibrate.harmonize_time(np.arange(3.0), t1[:3], t1[:3]), np.arange(3.0))
    assert calibrate.harmonize_time(np.array([[7.0]]), t1[:1], q).shape == (1,)


@pytest.mark.parametrize("kind", ["regular", "irregular"])
def test_mvbs_values_vs_brute_force(kind):
    # test_commongrid_api.py:363-436 (range_bin 2m, ping_time_bin 1s, atol=rtol=1e-10)
    d = kf.mock_small(kind)
    mv, t_left, r_left = commongrid.compute_MVBS(d["Sv"], d["echo_range"], d["ping_time"], "2m", "1s")
    exp = kf.brute_force_mvbs(d, "1s", 2)
    assert mv.shape == exp.shape
    np.testing.assert_allclose(mv, exp, atol=1e-10, rtol=1e-10, equal_nan=True)
    # NaN mask == histogram-of-coordinates mask (test _parse_nans)
    for c in range(2):
        for i, t0 in enumerate(t_left):
            t1 = t_left[i + 1] if i + 1 < len(t_left) else None
            sel = (d["ping_time"] >= t0) & ((d["ping_time"] <= t1) if t1 is not None else True)
            vals = d["echo_range"][c][sel]
            vals = vals[~np.isnan(vals)]
            hist, _ = np.histogram(vals, bins=np.append(r_left, r_left.max() + 2))
            np.testing.assert_array_equal(np.isnan(mv[c, i]), hist == 0)


def test_mvbs_shapes_regular_and_irregular():
    # test_commongrid_api.py:311-360
    d = kf.sv_regular()
    mv, _, _ = commongrid.compute_MVBS(d["Sv"], d["echo_range"], d["ping_time"], "5m", "10s")
    dt = (d["ping_time"][-1] - d["ping_time"][0]).astype("timedelta64[ns]").astype(np.int64)
    assert mv.shape == (2, int(np.ceil(dt / 1e9 / 10)), int(np.ceil(d["echo_range"].max() / 5)))
    d = kf.sv_irregular()
    mv, _, _ = commongrid.compute_MVBS(d["Sv"], d["echo_range"], d["ping_time"], "5m", "10s")
    def full_cols(a):
        return a[:, :, ~np.isnan(a).any(axis=(0, 1))].shape
    assert full_cols(mv[:, :3]) == (2, 3, 10)
    assert full_cols(mv[:, 3:12]) == (2, 9, 7)
    assert full_cols(mv[:, 12:]) == (2, 6, 3)


def test_mvbs_ping_time_is_resample_index():
    # test_commongrid_api.py:300-308
    d = kf.sv_regular(ping_time_interval="0.7s")
    _, t_left, _ = commongrid.compute_MVBS(d["Sv"], d["echo_range"], d["ping_time"], "20m", "20s")
    exp = pd.Series(0, index=pd.DatetimeIndex(d["ping_time"])).resample("20s").asfreq().index
    np.testing.assert_array_equal(t_left, exp.values)


@pytest.mark.parametrize("skipna,range_key", [(True, "depth"), (False, "depth"), (True, "echo_range"), (False, "echo_range")])
def test_mvbs_skipna_masks(skipna, range_key):
    # test_commongrid_api.py:471-556 -- first two pings of the irregular mock
    d = kf.mock_small("irregular")
    sub = {k: (v[:, :2] if v.ndim == 3 else v[:2]) for k, v in d.items()}
    mv, _, _ = commongrid.compute_MVBS(sub["Sv"], sub[range_key], sub["ping_time"], "2m", "20s", skipna=skipna)
    mask = np.isnan(mv)
    if range_key == "echo_range":
        exp = [[[False] * 5], [[False] * 5]]
    elif skipna:
        exp = [[[True, False, False, False, False, False]]] * 2
    else:
        exp = [[[True, True, True, False, False, True]], [[True, False, False, True, True, True]]]
    np.testing.assert_array_equal(mask, np.array(exp))


def test_mvbs_range_var_max_and_edges():
    d = kf.mock_small("regular")
    _, _, r_left = commongrid.compute_MVBS(d["Sv"], d["echo_range"], d["ping_time"], "1m", "20s", range_var_max="8m")
    assert r_left.max() == 8  # test_commongrid_api.py:580-592
    # arange(0, 10+5, 5) = [0,5,10] -> a sample at exactly 10 m is dropped
    e = commongrid.range_edges(np.array([0.0, 10.0]), 5.0)
    np.testing.assert_array_equal(e, [0, 5, 10])
    assert commongrid.bin_index(np.array([10.0]), e)[0] == -1
    assert commongrid.bin_index(np.array([10.0]), e, closed="right")[0] == 1


def test_mvbs_time_edges_midnight_anchored():
    t0 = np.datetime64("2018-07-01T13:47:07.300000000")
    pt = t0 + (np.arange(50) * 1e9).astype("timedelta64[ns]")
    assert commongrid.ping_edges(pt, "20s")[0] == np.datetime64("2018-07-01T13:47:00")
    assert commongrid.ping_edges(pt, "7s")[0] == np.datetime64("2018-07-01T13:47:03")


def test_mvbs_argument_errors():
    d = kf.mock_small("regular")
    with pytest.raises(TypeError):
        commongrid.compute_MVBS(d["Sv"], d["echo_range"], d["ping_time"], 10, "20s")
    with pytest.raises(ValueError, match="Range bin must be in meters"):
        commongrid.compute_MVBS(d["Sv"], d["echo_range"], d["ping_time"], "10km", "20s")
    with pytest.raises(TypeError, match="ping_time_bin must be a string"):
        commongrid.compute_MVBS(d["Sv"], d["echo_range"], d["ping_time"], "10m", 20)


def test_index_binning_matches_coarsen_formula():
    # test_commongrid_api.py:171-202: (4,100,4000), ping_num=3, range_sample_num=7
    d = kf.sv_regular(4, 4000, 0.5, 100)
    mv, er = commongrid.compute_MVBS_index_binning(d["Sv"], d["echo_range"], range_sample_num=7, ping_num=3)
    assert mv.shape == tuple(np.ceil((4, 100 / 3, 4000 / 7)).astype(int))
    # independent evaluation of one interior and one padded tail block
    lin = 10 ** (d["Sv"] / 10)
    np.testing.assert_allclose(mv[1, 2, 3], 10 * np.log10(lin[1, 6:9, 21:28].mean()), rtol=1e-14)
    np.testing.assert_allclose(mv[3, 33, 571], 10 * np.log10(lin[3, 99:, 3997:].mean()), rtol=1e-14)
    assert er[0, 0, 1] == d["echo_range"][0, 0, 7]


# ---- bin-edge membership pinned to pandas itself (round 4) ------------------------------------------------------------
@pytest.mark.parametrize("closed", ["left", "right"])
def test_bin_membership_equals_pandas_intervalindex(closed):
    """The reference bins through ``pd.IntervalIndex.from_breaks(edges, closed=closed)`` handed to flox as
    expected_groups (commongrid/utils.py:283-302, 592-627); flox factorizes against exactly those intervals.  flox cannot
    be installed here, pandas is: the oracle's ``bin_index`` (and through it every MVBS / NASC expectation and the HIP
    kernels held to them) must put every value -- values ON edges, just beside them, NaN, outside the grid -- where
    ``IntervalIndex.get_indexer`` and ``pd.cut`` put it.  Range edges as the reference builds them (np.arange(0, max +
    bin, bin), api.py:108-115) and time edges on a 20-s grid."""
    import pandas as pd

    from oracle import commongrid as ogrid

    rng = np.random.default_rng(5)
    for rbin in (1.0, 0.1, 0.07, 2.5):
        edges = np.arange(0, 37.3 + rbin, rbin)
        on = edges[rng.integers(0, len(edges), 200)]
        x = np.concatenate([on, np.nextafter(on, np.inf), np.nextafter(on, -np.inf), rng.uniform(-3, 45, 500),
                            [np.nan, -0.0, 0.0, edges[-1], edges[-1] + 1e-9, np.inf, -np.inf]])
        ii = pd.IntervalIndex.from_breaks(edges, closed=closed)
        want = ii.get_indexer(x)  # -1 outside / NaN
        np.testing.assert_array_equal(ogrid.bin_index(x, edges, closed), want)
        cut = pd.cut(x, bins=ii).codes
        np.testing.assert_array_equal(cut, want)
    t0 = np.datetime64("2026-05-01T00:00:00", "ns")
    t_edges = t0 + np.arange(0, 13) * np.timedelta64(20, "s")
    t = np.concatenate([t_edges, t_edges + np.timedelta64(1, "ns"), t_edges - np.timedelta64(1, "ns"),
                        t0 + rng.integers(-50, 300, 200) * np.timedelta64(1, "s"), [np.datetime64("NaT")]])
    ii = pd.IntervalIndex.from_breaks(pd.DatetimeIndex(t_edges), closed=closed)
    ok = ~np.isnat(t)
    want = np.full(t.shape, -1)
    want[ok] = ii.get_indexer(pd.DatetimeIndex(t[ok]))
    np.testing.assert_array_equal(ogrid.bin_index(t, t_edges, closed), want)
