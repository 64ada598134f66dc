"""Drop-in API parity of the noise masks and apply_mask on a real MI355X:
echopype_amd.clean.mask_* / echopype_amd.mask.apply_mask called like the reference's functions
(tests/clean/test_noise.py:41-150, 254-330, 445-540, 686-860; tests/mask/test_mask.py apply_mask
cases), compared with the CPU oracle on seeded synthetic scenes.
"""
import logging

import numpy as np
import pytest

from oracle import masks as omask
from test_gpu_masks import _scene

pytestmark = pytest.mark.gpu

DIMS = ("channel", "ping_time", "range_sample")


@pytest.fixture(scope="module")
def ep():
    import torch

    if not torch.cuda.is_available():
        pytest.fail("these tests need a GPU")
    import echopype_amd

    return echopype_amd


def _ds(ep, sv, depth, range_name="depth", on_device=False):
    C, P, S = sv.shape
    t0 = np.datetime64("2024-03-01T00:00:00", "ns")
    ds = ep.Dataset(coords={"channel": [f"ch_{i}" for i in range(C)],
                            "ping_time": t0 + np.arange(P) * np.timedelta64(1, "s"),
                            "range_sample": np.arange(S)})
    if on_device:
        import torch

        ds["Sv"] = ep.DataArray(ep.DeviceArray(torch.from_numpy(sv).cuda()), DIMS)
        ds[range_name] = ep.DataArray(ep.DeviceArray(torch.from_numpy(depth).cuda()), DIMS)
    else:
        ds["Sv"] = (DIMS, sv)
        ds[range_name] = (DIMS, depth)
    return ds


def _same_mask(got, exp, margin=None):
    got = np.asarray(got)
    assert got.dtype == np.bool_ and got.shape == exp.shape
    if margin is None:
        np.testing.assert_array_equal(got, exp)
    else:
        np.testing.assert_array_equal(got[margin], exp[margin])


# ---------------------------------------------------------------------------------- impulse noise
@pytest.mark.parametrize("use_index_binning", [False, True])
@pytest.mark.parametrize("on_device", [False, True])
def test_mask_impulse_noise(ep, use_index_binning, on_device):
    sv, depth = _scene(3, 40, 260, 21)
    ds = _ds(ep, sv, depth, on_device=on_device)
    m = ep.clean.mask_impulse_noise(ds, depth_bin="3m", num_side_pings=2, impulse_noise_threshold="10.0dB",
                                    use_index_binning=use_index_binning)
    assert m.dims == ("channel", "range_sample", "ping_time")  # apply_ufunc core dims go last
    exp = omask.mask_impulse_noise(sv, depth, "3m", 2, "10.0dB", use_index_binning)
    _same_mask(m.values, exp)
    assert exp.any() and not exp.all()
    # the reference's own property (tests/clean/test_noise.py:696-768): an unmasked interior sample has
    # at least one side difference <= threshold
    if use_index_binning:
        up = omask.index_binning_downsample_upsample(sv, depth, 3.0)
        got = m.values
        c, s, p = np.nonzero(~got[:, :, 2:-2])
        p = p + 2
        left, right = up[c, p, s] - up[c, p - 2, s], up[c, p, s] - up[c, p + 2, s]
        assert np.all((left <= 10.0) | (right <= 10.0))


def test_mask_impulse_noise_echo_range_f32(ep):
    sv, depth = _scene(2, 30, 200, 22)
    ds = _ds(ep, sv.astype(np.float32), depth.astype(np.float32), range_name="echo_range")
    m = ep.clean.mask_impulse_noise(ds, range_var="echo_range", use_index_binning=True).values
    exp = omask.mask_impulse_noise(sv.astype(np.float32).astype(np.float64),
                                   depth.astype(np.float32).astype(np.float64), use_index_binning=True)
    assert (m != exp).mean() < 2e-3  # fp32 decision noise only


# --------------------------------------------------------------------------------- transient noise
@pytest.mark.parametrize("func", ["nanmean", "nanmedian"])
@pytest.mark.parametrize("use_index_binning", [False, True])
def test_mask_transient_noise(ep, func, use_index_binning):
    C, P, S = 2, 16, 60
    sv, depth = _scene(C, P, S, 23, step=0.4)
    ds = _ds(ep, sv, depth)
    kw = dict(func=func, depth_bin="2m", num_side_pings=3, exclude_above="6.0m",
              transient_noise_threshold="8.0dB", use_index_binning=use_index_binning)
    m = ep.clean.mask_transient_noise(ds, **kw)
    assert m.dims == DIMS
    f = np.nanmean if func == "nanmean" else np.nanmedian
    pool = omask.index_binning_pool_Sv if use_index_binning else omask.pool_Sv
    pooled = pool(sv, depth, f, 2.0, 3, 6.0)
    with np.errstate(invalid="ignore"):
        margin = sv - pooled - 8.0
    exp = margin > 0
    _same_mask(m.values, exp, ~(np.abs(margin) < 1e-9))
    assert exp.any()
    # reference property (tests/clean/test_noise.py:264-329): unmasked & pooled finite -> Sv - pooled <= thr
    fin = np.isfinite(pooled) & ~m.values & np.isfinite(sv)
    assert np.all(sv[fin] - pooled[fin] <= 8.0 + 1e-9)
    if not use_index_binning:  # boundary pings / depths cannot be pooled (tests/clean/test_noise.py:161-219)
        assert np.isnan(pooled[:, :3]).all() and not m.values[:, :3].any()


def test_mask_transient_noise_value_window_ragged_and_f32(ep):
    sv, depth = _scene(2, 14, 50, 24, step=0.5, ragged=True)
    ds = _ds(ep, sv.astype(np.float32), depth.astype(np.float32))
    m = ep.clean.mask_transient_noise(ds, depth_bin="2m", num_side_pings=2, exclude_above="4.0m",
                                      transient_noise_threshold="6.0dB").values
    sv32, d32 = sv.astype(np.float32).astype(np.float64), depth.astype(np.float32).astype(np.float64)
    pooled = omask.pool_Sv(sv32, d32, np.nanmean, 2.0, 2, 4.0)
    with np.errstate(invalid="ignore"):
        margin = sv32 - pooled - 6.0
    _same_mask(m, margin > 0, ~(np.abs(margin) < 2e-3))


def test_mask_transient_noise_errors_and_warning(ep, caplog):
    sv, depth = _scene(1, 8, 20, 25)
    ds = _ds(ep, sv, depth)
    with pytest.raises(ValueError, match="Input `func` is `nanmode`. `func` must be `nanmean` or `nanmedian`."):
        ep.clean.mask_transient_noise(ds, func="nanmode")
    with caplog.at_level(logging.WARNING):
        ep.clean.mask_transient_noise(ds, func="nanmedian", depth_bin="0.2m", num_side_pings=2,
                                      exclude_above="250m")
    assert any("`func=nanmedian` is an incredibly slow operation" in r.message for r in caplog.records)
    bad = depth.copy()
    bad[0, 2, 5] = bad[0, 2, 4] - 1.0  # not monotone
    with pytest.raises(ValueError, match="non-decreasing"):
        ep.clean.mask_transient_noise(_ds(ep, sv, bad))


# ------------------------------------------------------------------------------- attenuated signal
@pytest.mark.parametrize("on_device", [False, True])
def test_mask_attenuated_signal(ep, on_device):
    sv, depth = _scene(3, 70, 300, 26, step=0.5, ragged=True)
    ds = _ds(ep, sv, depth, on_device=on_device)
    m = ep.clean.mask_attenuated_signal(ds, upper_limit_sl="30.0m", lower_limit_sl="90.0m", num_side_pings=6,
                                        attenuation_signal_threshold="-5.0dB")
    assert m.dims == DIMS
    exp = omask.mask_attenuated_signal(sv, depth, "30.0m", "90.0m", 6, "-5.0dB")
    _same_mask(m.values, exp)
    assert exp.any() and not exp.all()
    assert np.all(m.values.all(axis=2) | ~m.values.any(axis=2))  # whole pings


def test_mask_attenuated_signal_outside_searching_range_and_limit_error(ep):
    sv, depth = _scene(2, 20, 50, 27)
    ds = _ds(ep, sv, depth)
    m = ep.clean.mask_attenuated_signal(ds, upper_limit_sl="1800.0m", lower_limit_sl="2800.0m")
    assert not m.values.any() and m.values.shape == sv.shape  # tests/clean/test_noise.py:792-815
    with pytest.raises(ValueError, match="Minimum range has to be shorter than maximum range"):
        ep.clean.mask_attenuated_signal(ds, upper_limit_sl="180.0m", lower_limit_sl="170.0m")


@pytest.mark.parametrize("range_var", ["depth", "echo_range"])
def test_mask_functions_with_no_vertical_range_variables(ep, range_var):
    sv, depth = _scene(1, 8, 20, 28)
    ds = _ds(ep, sv, depth, range_name="echo_range" if range_var == "depth" else "depth")
    for fn in (ep.clean.mask_attenuated_signal, ep.clean.mask_impulse_noise, ep.clean.mask_transient_noise):
        with pytest.raises(ValueError):
            fn(ds, range_var=range_var)
    with pytest.raises(ValueError, match="`range_var` must be either `echo_range` or `depth`."):
        ep.clean.mask_impulse_noise(ds, range_var="range")


def test_mask_functions_dimensions_with_defaults(ep):
    """tests/clean/test_noise.py:73-93: default arguments on a small subset, coordinates carried over."""
    sv, depth = _scene(2, 6, 6, 29)
    ds = _ds(ep, sv, depth)
    for fn in (ep.clean.mask_attenuated_signal, ep.clean.mask_impulse_noise, ep.clean.mask_transient_noise):
        mask = fn(ds)
        assert set(mask.dims) == set(DIMS)
        for d in DIMS:
            np.testing.assert_array_equal(mask[d].values, ds[d].values)
            assert mask.sizes[d] == ds.sizes[d]


# -------------------------------------------------------------------------------------- apply_mask
def test_apply_mask_single_list_and_fill(ep):
    sv, depth = _scene(3, 25, 80, 30)
    ds = _ds(ep, sv, depth)
    ds.data_vars["Sv"].attrs["units"] = "dB"
    imp = ep.clean.mask_impulse_noise(ds, depth_bin="3m", use_index_binning=True)  # (C, S, P) dims
    att = ep.clean.mask_attenuated_signal(ds, upper_limit_sl="10.0m", lower_limit_sl="12.0m", num_side_pings=3,
                                          attenuation_signal_threshold="-4.0dB")
    rng = np.random.default_rng(1)
    host2d = ep.DataArray(rng.random((25, 80)) < 0.8, ("ping_time", "range_sample"))
    keep = [~imp.values.transpose(0, 2, 1), ~att.values, host2d.values]
    not_imp = ep.DataArray(~imp.values, imp.dims)
    not_att = ep.DataArray((~att.values).astype(np.float64), att.dims)  # numeric 0/1 mask is accepted

    out = ep.mask.apply_mask(ds, not_imp)
    np.testing.assert_array_equal(out["Sv"].values, omask.apply_mask(sv, keep[0]))
    assert out["Sv"].dims == DIMS
    np.testing.assert_array_equal(ds["Sv"].values, sv)  # the source dataset is untouched
    assert out["Sv"].attrs["long_name"] == "Volume backscattering strength, masked (Sv re 1 m-1)"
    assert out["Sv"].attrs["units"] == "dB"
    lo, hi = out["Sv"].attrs["actual_range"]
    assert lo == round(float(np.nanmin(out["Sv"].values)), 2) and hi == round(float(np.nanmax(out["Sv"].values)), 2)
    assert out.attrs["mask_function"] == "mask.apply_mask"

    out = ep.mask.apply_mask(ds, [host2d, not_imp, not_att], fill_value=-999)
    np.testing.assert_array_equal(out["Sv"].values, omask.apply_mask(sv, keep, -999.0))

    fill = ep.DataArray(rng.standard_normal((1, 25, 80)), DIMS)  # length-1 channel is squeezed out
    out = ep.mask.apply_mask(ds, host2d, fill_value=fill)
    np.testing.assert_array_equal(out["Sv"].values, np.where(host2d.values[None], sv, fill.values))


def test_apply_mask_other_variable_and_depth_dims(ep):
    """A mask on an MVBS-like grid with dims (ping_time, depth) applied to another variable name."""
    rng = np.random.default_rng(2)
    vals = rng.standard_normal((2, 9, 7)).astype(np.float32)
    ds = ep.Dataset(coords={"channel": ["a", "b"], "ping_time": np.arange(9).astype("datetime64[s]"),
                            "depth": np.arange(7.0)})
    ds["Sv_corrected"] = (("channel", "ping_time", "depth"), vals)
    m = ep.DataArray(rng.random((7, 9)) < 0.5, ("depth", "ping_time"))  # transposed on purpose
    out = ep.mask.apply_mask(ds, m, var_name="Sv_corrected")
    np.testing.assert_array_equal(out["Sv_corrected"].values, np.where(m.values.T[None], vals, np.nan))
    assert out["Sv_corrected"].dtype == np.float32


def test_apply_mask_errors(ep):
    sv, depth = _scene(2, 6, 8, 31)
    ds = _ds(ep, sv, depth)
    ok = ep.DataArray(np.ones((6, 8), bool), ("ping_time", "range_sample"))
    with pytest.raises(ValueError, match="Masks must have one of the following dimensions"):
        ep.mask.apply_mask(ds, ep.DataArray(np.ones((6, 8), bool), ("ping_time", "beam")))
    with pytest.raises(TypeError, match="Mask cannot contain NaN"):
        ep.mask.apply_mask(ds, ep.DataArray(np.full((6, 8), np.nan), ("ping_time", "range_sample")))
    with pytest.raises(TypeError, match="Mask must be boolean"):
        ep.mask.apply_mask(ds, ep.DataArray(np.full((6, 8), 2.0), ("ping_time", "range_sample")))
    with pytest.raises(ValueError, match="do not match the dimensions of source"):
        ep.mask.apply_mask(ds, ep.DataArray(np.ones((6, 8), bool), ("ping_time", "depth")))
    with pytest.raises(ValueError, match="not of the same shape"):
        ep.mask.apply_mask(ds, ep.DataArray(np.ones((6, 7), bool), ("ping_time", "range_sample")))
    with pytest.raises(ValueError, match="that dimension should match"):
        ep.mask.apply_mask(ds, ep.DataArray(np.ones((3, 6, 8), bool), DIMS))
    with pytest.raises(ValueError, match="does not contain the variable var_name"):
        ep.mask.apply_mask(ds, ok, var_name="TS")
    with pytest.raises(TypeError, match="var_name must be a string"):
        ep.mask.apply_mask(ds, ok, var_name=3)
    with pytest.raises(TypeError, match="fill_value must be of type"):
        ep.mask.apply_mask(ds, ok, fill_value="nan")
    with pytest.raises(ValueError, match="If fill_value is an array"):
        ep.mask.apply_mask(ds, ok, fill_value=ep.DataArray(np.zeros((6, 7)), ("ping_time", "range_sample")))
    with pytest.raises(ValueError, match="single dict because mask is a single value"):
        ep.mask.apply_mask(ds, ok, storage_options_mask=[{}])
    with pytest.raises(ValueError, match="same shape in the 'channel' dimension"):
        ep.mask.apply_mask(ds, [ep.DataArray(np.ones((2, 6, 8), bool), DIMS),
                                ep.DataArray(np.ones((2, 6, 7), bool), DIMS)])
    no_chan = ep.Dataset(coords={"ping_time": ds["ping_time"].values, "range_sample": np.arange(8)})
    no_chan["Sv"] = (("ping_time", "range_sample"), sv[0])
    with pytest.raises(ValueError, match="'channel' is a dimension in mask but not a dimension in source"):
        ep.mask.apply_mask(no_chan, ep.DataArray(np.ones((2, 6, 8), bool), DIMS))
    out = ep.mask.apply_mask(no_chan, ok)  # case 1 of the docstring: no channel anywhere
    np.testing.assert_array_equal(out["Sv"].values, sv[0])


# ------------------------------------------------------------------------------------- end to end
def test_calibrate_mask_apply_mvbs_chain(ep):
    """compute_Sv -> add_depth -> masks -> apply_mask -> compute_MVBS, everything device-resident."""
    d = ep.synth.ek60_numpy(2, 200, 600)
    ds = ep.calibrate.compute_Sv(ep.echodata.from_ek60_arrays(d))
    ds = ep.consolidate.add_depth(ds, depth_offset=5.0)
    masks = [ep.clean.mask_impulse_noise(ds, depth_bin="2m", use_index_binning=True),
             ep.clean.mask_transient_noise(ds, depth_bin="2m", num_side_pings=4, exclude_above="10.0m",
                                           use_index_binning=True)]
    sv, depth = ds["Sv"].values, ds["depth"].values
    exp_masks = [omask.mask_impulse_noise(sv, depth, "2m", use_index_binning=True).transpose(0, 2, 1),
                 omask.mask_transient_noise(sv, depth, depth_bin="2m", num_side_pings=4, exclude_above="10.0m",
                                            use_index_binning=True)]
    keep = [ep.DataArray(~m.values, m.dims) for m in masks]
    clean = ep.mask.apply_mask(ds, keep)
    exp = omask.apply_mask(sv, [~m for m in exp_masks])
    agree = np.isnan(clean["Sv"].values) == np.isnan(exp)
    assert agree.mean() > 0.9999  # threshold-margin flips only
    mvbs = ep.commongrid.compute_MVBS(clean, range_var="depth", range_bin="5m", ping_time_bin="20s")
    assert np.isfinite(mvbs["Sv"].values).any()


# ------------------------------------------------- the reference's own functions, executed end to end
@pytest.mark.parametrize("tag", ["tri0", "tri1", "tri2", "trv0", "trv1", "imp0", "imp1", "imp2", "att0", "att1", "att2"])
def test_mask_api_vs_reference_goldens(ep, tag):
    """tests/golden/ref_maskapi_goldens.npz holds masks produced by the reference's clean/api.py functions
    themselves (oracle/gen_maskapi_goldens.py); same call, same keyword strings, same dimension order."""
    import os

    from test_oracle_masks import API_GOLDEN, _kw

    assert os.path.exists(API_GOLDEN)
    gold = np.load(API_GOLDEN)
    sv, er, kw = gold[f"{tag}_Sv"], gold[f"{tag}_echo_range"], _kw(gold, tag)
    ds = _ds(ep, sv, er, range_name="echo_range")
    if tag.startswith("tr"):
        m = ep.clean.mask_transient_noise(ds, range_var="echo_range", use_index_binning=tag.startswith("tri"), **kw)
    elif tag.startswith("imp"):
        m = ep.clean.mask_impulse_noise(ds, range_var="echo_range", use_index_binning=True, **kw)
    else:
        m = ep.clean.mask_attenuated_signal(ds, range_var="echo_range", **kw)
    assert list(m.dims) == gold[f"{tag}_dims"].tolist()
    _same_mask(m.values, gold[f"{tag}_mask"])
