"""The array shim the reference's code is executed over (oracle/xr_shim.py) stands in for xarray, which this image
cannot install.  These tests hold every labelled-array rule the golden generators lean on to an INDEPENDENT
implementation of the same rule: pandas (the library xarray itself delegates label handling to: Index.slice_indexer,
Grouper / resample, reindex) or a plain NumPy loop written from xarray's documented behaviour.  They are the check
that the shim's answers are xarray's semantics and not a reading of them: broadcasting by dimension NAME, inner-join
alignment, NaN-skipping reductions, coarsen with boundary="pad" (all window axes reduced together, mean labels),
label slices with both ends inclusive, forward-fill reindex, diff labels, where / fillna, apply_ufunc core dims."""
import numpy as np
import pandas as pd
import pytest

from oracle import xr_shim as xr


def _da(a, **dims):
    return xr.DataArray(a, coords={k: v for k, v in dims.items()}, dims=list(dims))


def test_binary_ops_broadcast_by_dimension_name_not_position():
    rng = np.random.default_rng(0)
    a = _da(rng.normal(size=(3, 4)), channel=np.arange(3), ping_time=np.arange(4))
    b = _da(rng.normal(size=(4, 5)), ping_time=np.arange(4), range_sample=np.arange(5))
    got = a * b
    assert got.dims == ("channel", "ping_time", "range_sample")  # order of first appearance
    np.testing.assert_array_equal(got.values, a.values[:, :, None] * b.values[None, :, :])
    # transposed operand: names decide, not positions
    bt = b.transpose("range_sample", "ping_time")
    np.testing.assert_array_equal((a * bt).values, got.values)
    # a NumPy scalar / 0-d on either side keeps dims; the reflexive form keeps the DataArray's order
    np.testing.assert_array_equal((2.0 - a).values, 2.0 - a.values)
    assert (b - a).dims == ("ping_time", "range_sample", "channel")


def test_arithmetic_aligns_shared_dimension_by_inner_join_like_pandas():
    """xarray arithmetic joins differing labels of a shared dimension with join="inner"; pandas joins outer and leaves
    NaN where a label is missing -- dropping those NaN rows gives the inner join."""
    a = _da(np.array([1.0, 2.0, 3.0, 4.0]), channel=np.array(["c1", "c2", "c3", "c4"]))
    b = _da(np.array([10.0, 20.0]), channel=np.array(["c4", "c2"]))
    got = a + b
    want = (pd.Series(a.values, index=a.coords["channel"]) + pd.Series(b.values, index=b.coords["channel"])).dropna()
    assert list(got.coords["channel"]) == ["c2", "c4"]  # the order of the left operand
    np.testing.assert_array_equal(got.values, want.loc[["c2", "c4"]].values)


@pytest.mark.parametrize("skipna", [True, False])
def test_reductions_skip_nan_like_pandas(skipna):
    rng = np.random.default_rng(1)
    x = rng.normal(size=(6, 7))
    x[rng.random(x.shape) < 0.3] = np.nan
    x[2] = np.nan  # an all-NaN row: NaN result either way
    da = _da(x, ping_time=np.arange(6), range_sample=np.arange(7))
    df = pd.DataFrame(x)
    np.testing.assert_allclose(da.min(dim="range_sample", skipna=skipna).values, df.min(axis=1, skipna=skipna).values)
    np.testing.assert_allclose(da.max(dim="range_sample", skipna=skipna).values, df.max(axis=1, skipna=skipna).values)
    np.testing.assert_allclose(da.mean(dim="ping_time", skipna=skipna).values, df.mean(axis=0, skipna=skipna).values)
    if skipna:  # the default for float AND complex data (the sector average of calibrate_ek.py:483 is a complex mean)
        np.testing.assert_allclose(da.mean(dim="ping_time").values, df.mean(axis=0, skipna=True).values)
        np.testing.assert_allclose(float(da.mean()), np.nanmean(x))
        z = _da(x + 1j * x[::-1], ping_time=np.arange(6), range_sample=np.arange(7))
        np.testing.assert_allclose(z.mean(dim="range_sample").values, np.nanmean(z.values, axis=1))
        np.testing.assert_array_equal(_da(np.arange(6), ping_time=np.arange(6)).mean().values, 2.5)  # ints: plain mean


@pytest.mark.parametrize("P,S,pn,sn", [(10, 9, 3, 4), (7, 12, 7, 5), (5, 5, 1, 1), (4, 6, 10, 10)])
def test_coarsen_pad_reduces_all_window_axes_together_and_labels_by_mean(P, S, pn, sn):
    """coarsen(ping_time=pn, range_sample=sn, boundary="pad"): trailing partial windows are kept (NaN padding), the
    reduction runs over the whole 2-D block at once -- NOT a mean of per-row means, which weights rows with fewer valid
    samples differently -- and the new coordinate of a coarsened dimension is the mean of the labels in the window
    (coord_func="mean").  Independent form: a groupby of the flattened samples on the block ids, in pandas."""
    rng = np.random.default_rng(P * 100 + S)
    x = rng.normal(size=(P, S))
    x[rng.random(x.shape) < 0.25] = np.nan
    t = np.datetime64("2026-01-01T00:00:00", "ns") + (np.arange(P) * 1_500_000_000).astype("timedelta64[ns]")
    da = _da(x, ping_time=t, range_sample=np.arange(S))
    got = da.coarsen(ping_time=pn, range_sample=sn, boundary="pad").mean()
    got_min = da.coarsen(ping_time=pn, range_sample=sn, boundary="pad").min()
    pi, si = np.meshgrid(np.arange(P) // pn, np.arange(S) // sn, indexing="ij")
    flat = pd.DataFrame({"p": pi.ravel(), "s": si.ravel(), "v": x.ravel()})
    want = flat.groupby(["p", "s"])["v"].mean().unstack("s").values      # pandas means skip NaN; all-NaN block -> NaN
    want_min = flat.groupby(["p", "s"])["v"].min().unstack("s").values
    np.testing.assert_allclose(got.values, want, rtol=1e-13, atol=0)  # (summation order)
    np.testing.assert_allclose(got_min.values, want_min)
    # labels: mean per window -- integers become floats, datetimes the mean instant
    np.testing.assert_allclose(got.coords["range_sample"], pd.Series(np.arange(S, dtype=float)).groupby(np.arange(S) // sn).mean().values)
    want_t = pd.Series(t.astype("int64").astype(float)).groupby(np.arange(P) // pn).mean().values
    np.testing.assert_allclose(got.coords["ping_time"].astype("int64").astype(float), want_t, rtol=0, atol=1.0)  # whole ns


def test_label_slice_is_inclusive_at_both_ends_like_pandas_loc():
    r = np.array([0.0, 0.5, 1.0, 1.5, 2.0, 2.5])
    da = _da(np.arange(6.0), echo_range=r)
    s = pd.Series(np.arange(6.0), index=r)
    for lo, hi in [(0.5, 2.0), (0.4, 2.1), (1.0, 1.0), (2.6, 9.0), (-1.0, 0.0)]:
        np.testing.assert_array_equal(da.sel(echo_range=slice(lo, hi)).values, s.loc[lo:hi].values)
    t = np.datetime64("2026-01-01", "ns") + (np.arange(5) * 10**9).astype("timedelta64[ns]")
    dt = _da(np.arange(5.0), ping_time=t)
    st = pd.Series(np.arange(5.0), index=t)
    np.testing.assert_array_equal(dt.sel(ping_time=slice(t[1], t[3])).values, st.loc[t[1]:t[3]].values)


@pytest.mark.parametrize("freq", ["2s", "20s", "1min"])
def test_resample_labels_are_pandas_bins(freq):
    rng = np.random.default_rng(3)
    t = np.datetime64("2026-05-01T23:59:31", "ns") + (np.cumsum(rng.uniform(0.2, 9.0, 40)) * 1e9).astype("timedelta64[ns]")
    da = _da(np.arange(40.0), ping_time=t)
    got = da.resample(ping_time=freq, skipna=True).first().indexes["ping_time"]
    want = pd.Series(np.arange(40.0), index=t).resample(freq).first().index
    np.testing.assert_array_equal(np.asarray(got), np.asarray(want))


def test_reindex_ffill_and_reindex_like_match_pandas():
    src = _da(np.array([[1.0, 2.0, 3.0], [4.0, 5.0, 6.0]]), channel=np.arange(2), ping_time=np.array([0, 20, 40]))
    tgt = np.array([-5, 0, 1, 19, 20, 39, 40, 100])
    got = src.reindex({"ping_time": tgt}, method="ffill")
    want = pd.DataFrame(src.values.T, index=[0, 20, 40]).reindex(tgt, method="ffill").values.T
    np.testing.assert_array_equal(got.values, want)  # labels before the first source label -> NaN
    other = _da(np.zeros((2, 4)), channel=np.arange(2), ping_time=np.array([20, 30, 40, 0]))
    got2 = src.reindex_like(other)
    want2 = pd.DataFrame(src.values.T, index=[0, 20, 40]).reindex([20, 30, 40, 0]).values.T
    np.testing.assert_array_equal(got2.values, want2)


def test_diff_where_fillna_isnull_match_pandas():
    x = np.array([3.0, np.nan, 7.0, 8.5, np.nan, 1.0])
    da = _da(x, ping_time=np.arange(6) * 10)
    s = pd.Series(x, index=np.arange(6) * 10)
    d = da.diff("ping_time")  # label="upper": the result carries the labels of the later elements
    np.testing.assert_array_equal(d.values, s.diff().iloc[1:].values)
    np.testing.assert_array_equal(d.coords["ping_time"], s.index[1:])
    np.testing.assert_array_equal(da.where(da > 5).values, s.where(s > 5).values)          # NaN compares False
    np.testing.assert_array_equal(da.where(da > 5, -1.0).values, s.where(s > 5, -1.0).values)
    np.testing.assert_array_equal(da.fillna(0.0).values, s.fillna(0.0).values)
    np.testing.assert_array_equal(da.isnull().values, s.isnull().values)
    np.testing.assert_array_equal(xr.where(da > 5, da, 0.0).values, np.where(x > 5, x, 0.0))


def test_apply_ufunc_moves_core_dims_last_and_loops_over_the_rest():
    rng = np.random.default_rng(5)
    a = _da(rng.normal(size=(2, 3, 8)), channel=np.arange(2), ping_time=np.arange(3), range_sample=np.arange(8))
    k = np.array([0.25, 0.5, 0.25])

    def smooth(v):  # 1-D in, 1-D out: vectorize=True calls it once per (channel, ping)
        assert v.ndim == 1
        return np.convolve(v, k, mode="same")

    got = xr.apply_ufunc(smooth, a, input_core_dims=[["range_sample"]], output_core_dims=[["range_sample"]], vectorize=True)
    want = np.stack([[np.convolve(a.values[c, p], k, mode="same") for p in range(3)] for c in range(2)])
    np.testing.assert_array_equal(got.transpose("channel", "ping_time", "range_sample").values, want)
    # a core dim that is not last in the input is moved last before the call
    at = a.transpose("range_sample", "channel", "ping_time")
    got_t = xr.apply_ufunc(smooth, at, input_core_dims=[["range_sample"]], output_core_dims=[["range_sample"]], vectorize=True)
    np.testing.assert_array_equal(got_t.transpose("channel", "ping_time", "range_sample").values, want)


def test_scalar_conversions_and_strictness():
    one = _da(np.array([2.5]), channel=np.arange(1))
    assert float(one) == 2.5 and int(_da(np.array([3]), channel=np.arange(1))) == 3
    with pytest.raises((TypeError, ValueError)):  # more than one element: no implicit scalar
        float(_da(np.array([1.0, 2.0]), channel=np.arange(2)))
    a = _da(np.zeros((2, 3)), channel=np.arange(2), ping_time=np.arange(3))
    with pytest.raises(TypeError):  # an unlabelled ndarray of another shape does not broadcast silently
        a + np.zeros(3)
    # equal lengths, PERMUTED labels: joined on the labels in the first operand's order (xarray's default inner join =
    # pandas Index.intersection, executed by the shim) ...
    x = _da(np.array([1.0, 2.0, 3.0]), channel=np.array(["a", "b", "c"]))
    y = _da(np.array([10.0, 20.0, 30.0]), channel=np.array(["b", "a", "c"]))
    got = x + y
    assert list(got.coords["channel"]) == list(pd.Index(["a", "b", "c"]).intersection(pd.Index(["b", "a", "c"])))
    np.testing.assert_array_equal(got.values, (pd.Series([1.0, 2, 3], index=list("abc"))
                                              + pd.Series([10.0, 20, 30], index=list("bac")))[list(got.coords["channel"])])
    got = y - x
    assert list(got.coords["channel"]) == ["b", "a", "c"]
    np.testing.assert_array_equal(got.values, [8.0, 19.0, 27.0])
    with pytest.raises(AssertionError):  # ... OTHER labels (the join would drop some): refused
        x + _da(np.arange(3.0), channel=np.array(["a", "b", "d"]))
    with pytest.raises((AssertionError, KeyError)):  # same dimension, different lengths, no labels to join on
        a + xr.DataArray(np.zeros(4), dims=["ping_time"])


def test_idxmin_sortby_expand_dims_and_pointwise_sel_against_pandas():
    """The methods calibrate/cal_params.py::get_vend_cal_params_power chains (idxmin over pulse_length_bin, sortby,
    expand_dims, sel with a labelled indexer), each against pandas / NumPy on the same numbers."""
    rng = np.random.default_rng(3)
    a = rng.normal(size=(4, 3, 5))
    a[1, 2, :] = np.nan                      # an all-NaN slice
    a[0, 0, 1] = np.nan
    da = _da(a, ping_time=np.arange(4), channel=np.array(["x", "y", "z"]), plb=np.array([10, 11, 12, 13, 14]))
    got = da.idxmin(dim="plb")
    exp = pd.DataFrame(a.reshape(12, 5), columns=[10, 11, 12, 13, 14]).idxmin(axis=1, skipna=True).to_numpy(dtype=float)
    assert got.dims == ("ping_time", "channel")
    np.testing.assert_array_equal(got.values, exp.reshape(4, 3))
    # sortby a coordinate, descending == pandas sort of the labels
    s = da.sortby(da["channel"], ascending=False)
    order = pd.Series(np.arange(3), index=["x", "y", "z"]).sort_index(ascending=False)
    assert list(s.coords["channel"]) == list(order.index)
    np.testing.assert_array_equal(s.values, a[:, order.to_numpy(), :])
    # expand_dims(name=labels): a new leading dimension, data repeated
    e = da.isel(ping_time=0).expand_dims(t=np.array([7, 8]))
    assert e.dims == ("t", "channel", "plb") and list(e.coords["t"]) == [7, 8]
    np.testing.assert_array_equal(e.values, np.broadcast_to(a[0][None], (2, 3, 5)))
    # pointwise selection by label along plb with an indexer on (ping_time, channel)
    idx = _da(rng.choice([10, 11, 12, 13, 14], size=(4, 3)), ping_time=np.arange(4), channel=np.array(["x", "y", "z"]))
    got = da.transpose("plb", "ping_time", "channel").sel(plb=idx, drop=True)
    assert got.dims == ("ping_time", "channel")
    exp = np.take_along_axis(a, (idx.values - 10)[..., None], axis=2)[..., 0]
    np.testing.assert_array_equal(got.values, exp)


def test_dropna_squeeze_and_linear_interp_with_extrapolation():
    """What calibrate/env_params.py::harmonize_env_param_time and utils/align.py::align_to_ping_time use: dropna along a
    dimension, squeeze(dim), interp(method="linear", fill_value="extrapolate") on datetime labels -- against NumPy."""
    t = np.datetime64("2017-06-20T01:00:00", "ns") + np.arange(5) * np.timedelta64(30, "s")
    v = np.array([0.0, 1.0, np.nan, 3.0, 5.0])
    da = xr.DataArray(v, coords={"time1": t}, dims=["time1"])
    d = da.dropna(dim="time1")
    np.testing.assert_array_equal(d.values, v[~np.isnan(v)])
    np.testing.assert_array_equal(d.coords["time1"], t[~np.isnan(v)])
    one = xr.DataArray(np.array([[7.0], [8.0]]), coords={"channel": np.arange(2), "time1": t[:1]}, dims=["channel", "time1"])
    sq = one.squeeze(dim="time1")
    assert sq.dims == ("channel",) and "time1" in sq.coords and sq.drop_vars("time1").dims == ("channel",)
    q = np.array(["2017-06-20T01:00:15", "2017-06-20T00:59:30", "2017-06-20T01:03:00"], dtype="datetime64[ns]")
    got = d.interp({"time1": xr.DataArray(q, coords={"ping_time": q}, dims=["ping_time"])}, method="linear",
                   kwargs={"fill_value": "extrapolate"})
    x = (d.coords["time1"] - t[0]).astype(np.int64).astype(float)
    xq = (q - t[0]).astype(np.int64).astype(float)
    y = d.values
    exp = np.interp(xq, x, y)
    exp[1] = y[0] + (y[1] - y[0]) * (xq[1] - x[0]) / (x[1] - x[0])       # linear extrapolation on either side
    exp[2] = y[-1] + (y[-1] - y[-2]) * (xq[2] - x[-1]) / (x[-1] - x[-2])
    assert got.dims == ("ping_time",)
    np.testing.assert_allclose(got.values, exp, rtol=1e-14)
