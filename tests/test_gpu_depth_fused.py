"""consolidate.add_depth fused into the Sv -> MVBS pass: compute_Sv -> add_depth ->
compute_MVBS(range_var="depth") is ONE sweep of the raw power samples (epa_sv_mvbs_fused_depth, 12 B/sample) -- the
depth array is an affine function of the coefficient rows and is written only when somebody reads it.

Reference: consolidate/api.py:68-243 (core :221), commongrid/api.py:30-191; the oracle restates both.
"""
import logging

import numpy as np
import pytest

import oracle_chain as oc
from oracle import commongrid as ogrid

pytestmark = pytest.mark.gpu

SAMPLE_KERNELS = ("fused_sv_mvbs_kernel", "sv_power_kernel", "sv_power_piece_kernel", "depth_rows_kernel",
                  "block_reduce_kernel", "mvbs_of_sv_fixed_kernel", "mvbs_of_sv_rows_kernel", "range_power_kernel")


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


def _case(ep, C=3, P=240, S=1000, ss_every=7):
    d = ep.synth.ek60_numpy(C, P, S, ss_every=ss_every)
    d["backscatter_r"][1, 5, S - 60:] = np.nan  # a NaN-padded ping: echo_range, hence depth, is NaN there
    d["backscatter_r"][0, 11, 3] = np.nan
    return d


def _sample_kernels(tr):
    return [k for k in tr.kernels if k in SAMPLE_KERNELS]


@pytest.mark.parametrize("dtype,rtol", [("float64", 1e-9), ("float32", 1e-3)])
@pytest.mark.parametrize("downward", [True, False])
def test_three_calls_one_pass(ep, dtype, rtol, downward):
    """The three reference calls launch ONE kernel that touches the samples; Sv is bit-identical to the eager K1 array,
    the MVBS is the oracle's on the oracle's depth, the depth statistics are the written array's, and depth stays
    unwritten until it is read -- then it holds the reference's bits (two roundings, consolidate/api.py:221)."""
    from echopype_amd import _lib

    d = _case(ep)
    P = d["backscatter_r"].shape[1]
    tilt = ep.DataArray(np.linspace(0.0, 20.0, P), ("ping_time",), coords={"ping_time": d["ping_time"]})
    off = ep.DataArray(np.linspace(3.0, 4.0, P) + (0.0 if downward else 400.0), ("ping_time",),
                       coords={"ping_time": d["ping_time"]})
    ed = ep.echodata.from_ek60_arrays(d)
    with _lib.launch_trace() as tr:
        ds = ep.calibrate.compute_Sv(ed, dtype=dtype)
        ep.consolidate.add_depth(ds, depth_offset=off, tilt=tilt, downward=downward)
        mv = ep.commongrid.compute_MVBS(ds, range_var="depth", range_bin="2m", ping_time_bin="10s")
        mv["Sv"].values
    if dtype == "float64":
        assert _sample_kernels(tr) == ["fused_sv_mvbs_kernel"], tr.kernels
        assert ds["Sv"].data.materialized and not ds["depth"].data.materialized and not ds["echo_range"].data.materialized
    else:  # (a float32 Sv is never left deferred, calibrate_ek.py::_cal_power_samples: K1, then depth is written and binned)
        assert "fused_sv_mvbs_kernel" not in tr.kernels

    sv, er = oc.ek60(d, "Sv")
    scale = (1.0 if downward else -1.0) * np.cos(np.deg2rad(tilt.values))
    if dtype == "float32":
        er_t = er.astype(np.float32)
        exp_depth = off.values.astype(np.float32)[None, :, None] + scale.astype(np.float32)[None, :, None] * er_t
    else:
        exp_depth = off.values[None, :, None] + scale[None, :, None] * er
    exp, t_left, r_left = ogrid.compute_MVBS(sv, exp_depth.astype(np.float64), d["ping_time"], "2m", "10s")
    if dtype == "float64":
        close(mv["Sv"].values, exp, rtol, "MVBS on the lazy depth")
    else:  # (float32 Sv values move a sample in and out of a bin's mean by 1e-3 relative: compare where the bins agree)
        got = mv["Sv"].values
        assert got.shape == exp.shape
        ok = np.isfinite(got) & np.isfinite(exp)
        assert ok.mean() > 0.9 * np.isfinite(exp).mean()
        np.testing.assert_allclose(got[ok], exp[ok], rtol=rtol, atol=2e-2)
    np.testing.assert_array_equal(mv["depth"].values, r_left)
    np.testing.assert_array_equal(mv["ping_time"].values, t_left)

    # the eager pieces: K1's Sv (bit for bit), the depth array (the reference's bits for float64)
    ds2 = ep.calibrate.compute_Sv(ed, dtype=dtype)
    np.testing.assert_array_equal(ds["Sv"].values, ds2["Sv"].values)
    st = ds["depth"].data.cached_stats()
    assert dtype == "float32" or not ds["depth"].data.materialized
    got_depth = ds["depth"].values
    assert ds["depth"].data.materialized
    if dtype == "float64":
        np.testing.assert_array_equal(np.isnan(got_depth), np.isnan(exp_depth))
        fin = np.isfinite(exp_depth)
        np.testing.assert_array_equal(got_depth[fin], exp_depth[fin])  # the reference's bits: product rounded, then the sum
    assert st == (np.nanmin(got_depth), np.nanmax(got_depth), int(np.isnan(got_depth).sum())) and st[2] > 0


def test_fused_depth_equals_binning_the_written_depth(ep):
    """Same dataset, depth written first (an array: the generic binning kernel on Sv + depth): the same MVBS up to the
    order of a bin's additions, the same grid -- per-channel beam-angle scaling with a NaN channel (a zero direction
    vector, ek_depth_utils.py:80-112) included: its bins stay empty and its depth counts as NaN coordinates."""
    from echopype_amd import _lib

    d = _case(ep, C=3, P=120, S=800)
    ed = ep.echodata.from_ek60_arrays(d)
    C, P = 3, 120
    scale = ep.DataArray(np.array([1.0, 0.8, np.nan]), ("channel",))
    results = []
    for written in (False, True):
        ds = ep.calibrate.compute_Sv(ed)
        # (add_depth's own options go through echodata; the per-channel scaling is injected the way ek_use_beam_angles
        # returns it)
        import echopype_amd.consolidate.api as capi

        orig = capi.ek_use_beam_angles
        capi.ek_use_beam_angles = lambda beam: (scale.values, ("channel",))
        try:
            fed = _FakeEchodata(ed)
            ep.consolidate.add_depth(ds, fed, depth_offset=1.5, use_beam_angles=True)
        finally:
            capi.ek_use_beam_angles = orig
        if written:
            ds["depth"] = ep.DataArray(ep.DeviceArray(ds["depth"].data.tensor.clone()), ds["depth"].dims)
        with _lib.launch_trace() as tr:
            mv = ep.commongrid.compute_MVBS(ds, range_var="depth", range_bin="1m", ping_time_bin="20s")
            vals = mv["Sv"].values
        results.append((vals, mv["depth"].values, _sample_kernels(tr)))
    (a, ga, ka), (b, gb, kb) = results
    assert ka == ["fused_sv_mvbs_kernel"] and "fused_sv_mvbs_kernel" not in kb, (ka, kb)
    np.testing.assert_array_equal(ga, gb)
    np.testing.assert_array_equal(np.isnan(a), np.isnan(b))
    assert np.isnan(a[2]).all() and np.isfinite(a[:2]).any()
    np.testing.assert_allclose(a, b, rtol=1e-12, atol=1e-12)


class _FakeEchodata:
    """An EchoData whose Sonar group says EK60 and whose beam group carries the channel coordinate (all add_depth's
    ``use_beam_angles`` branch looks at before it calls ek_use_beam_angles)."""

    def __init__(self, ed):
        self._ed = ed
        self.sonar_model = "EK60"

    def __getitem__(self, key):
        return self._ed[key]

    def __bool__(self):
        return True


def test_scalar_offset_and_range_var_max(ep):
    """Numbers for depth_offset / tilt (nothing is uploaded), ``range_var_max`` given: the grid is the caller's."""
    d = _case(ep, C=2, P=100, S=600, ss_every=1000)
    ds = ep.calibrate.compute_Sv(ep.echodata.from_ek60_arrays(d))
    ep.consolidate.add_depth(ds, depth_offset=2.5, tilt=15.0)
    mv = ep.commongrid.compute_MVBS(ds, range_var="depth", range_bin="2m", ping_time_bin="20s", range_var_max="50m")
    sv, er = oc.ek60(d, "Sv")
    exp_depth = 2.5 + er * np.cos(np.deg2rad(15.0))
    exp, _, r_left = ogrid.compute_MVBS(sv, exp_depth, d["ping_time"], "2m", "20s", range_var_max="50m")
    close(mv["Sv"].values, exp, 1e-9, "MVBS on depth, range_var_max")
    np.testing.assert_array_equal(mv["depth"].values, r_left)
    assert not ds["depth"].data.materialized


def test_depth_edited_in_place_takes_the_plain_route(ep, caplog):
    """A depth somebody wrote to after reading it is no longer the affine function: binned as the array it is."""
    from echopype_amd import _lib

    d = _case(ep, C=2, P=100, S=600)
    ds = ep.calibrate.compute_Sv(ep.echodata.from_ek60_arrays(d))
    ep.consolidate.add_depth(ds, depth_offset=1.0)
    ds["depth"].data.tensor.mul_(2.0)
    with _lib.launch_trace() as tr:
        mv = ep.commongrid.compute_MVBS(ds, range_var="depth", range_bin="5m", ping_time_bin="20s")
        got = mv["Sv"].values
    assert "fused_sv_mvbs_kernel" not in tr.kernels or "depth_rows_kernel" not in _sample_kernels(tr)
    sv, er = oc.ek60(d, "Sv")
    exp, _, r_left = ogrid.compute_MVBS(sv, 2.0 * (1.0 + er), d["ping_time"], "5m", "20s")
    close(got, exp, 1e-9, "MVBS on the edited depth")
    np.testing.assert_array_equal(mv["depth"].values, r_left)


def test_ops_level_depth_written_by_the_same_pass(ep):
    """epa_sv_mvbs_fused_depth with depth_out: the array the pass writes is epa_depth_rows' array bit for bit (NaN where
    the raw sample is), next to the same Sv and bins."""
    import torch
    from echopype_amd import ops

    d = _case(ep, C=2, P=80, S=1000)
    ed = ep.echodata.from_ek60_arrays(d)
    for dtype in ("float64", "float32"):
        cal = ep.calibrate.api.CALIBRATOR["EK60"](ed, None, None, None, dtype=dtype)
        raw, coef, flags, _ = cal._power_inputs("Sv")
        C, P, S = raw.shape
        g = torch.Generator().manual_seed(5)
        scale = (0.7 + 0.3 * torch.rand((C, P), generator=g, dtype=torch.float64)).cuda()
        scale[1, 40:] *= -1.0
        offset = (5.0 + torch.rand((C, P), generator=g, dtype=torch.float64)).cuda()
        offset[1, 40:] += 300.0
        ns = d["ping_time"].astype("datetime64[ns]").astype(np.int64)
        e0, dt = ns[0], 20 * 10**9
        n_t = int((ns[-1] - e0) // dt) + 1
        bin_start = ops.time_bin_offsets(ops.to_device(ns), int(e0), int(dt), n_t)
        tdt = getattr(torch, dtype)
        a = ops.sv_mvbs_fused_depth(raw, coef, scale, offset, bin_start, n_t, 2.0, 200, dtype=tdt, want_depth=True)
        b = ops.sv_mvbs_fused_depth(raw, coef, scale, offset, bin_start, n_t, 2.0, 200, dtype=tdt)
        dep, st = ops.depth_rows(scale, offset, coef=coef, mask_raw=raw, shape=(C, P, S), dtype=tdt)
        np.testing.assert_array_equal(a["depth"].cpu().numpy(), dep.cpu().numpy())
        np.testing.assert_array_equal(a["Sv"].cpu().numpy(), b["Sv"].cpu().numpy())
        np.testing.assert_array_equal(a["range_stats"].cpu().numpy(), st.cpu().numpy())
        np.testing.assert_array_equal(b["range_stats"].cpu().numpy(), st.cpu().numpy())
        np.testing.assert_allclose(a["MVBS"].cpu().numpy(), b["MVBS"].cpu().numpy(), rtol=1e-12 if dtype == "float64" else 1e-5,
                                   atol=1e-12 if dtype == "float64" else 1e-4)
        # ... and the bins are those of the generic kernel on the written arrays
        res = ops.mvbs(a["Sv"], bin_start, n_t, 2.0, 200, range=dep)
        np.testing.assert_array_equal(np.isnan(res["MVBS"].cpu().numpy()), np.isnan(a["MVBS"].cpu().numpy()))
        np.testing.assert_allclose(a["MVBS"].cpu().numpy(), res["MVBS"].cpu().numpy(), rtol=1e-12 if dtype == "float64" else 1e-4,
                                   atol=1e-12 if dtype == "float64" else 1e-3)


def test_depth_written_by_the_binning_pass_on_request(ep, monkeypatch):
    """EPA_DEPTH_WITH_MVBS=1: compute_MVBS(range_var="depth") writes the depth array next to Sv and the bins -- still one
    kernel over the samples, the array the lazy route would have produced."""
    from echopype_amd import _lib

    d = _case(ep, C=2, P=100, S=600)
    ed = ep.echodata.from_ek60_arrays(d)
    ref = ep.calibrate.compute_Sv(ed)
    ep.consolidate.add_depth(ref, depth_offset=3.0, tilt=10.0)
    mv0 = ep.commongrid.compute_MVBS(ref, range_var="depth", range_bin="2m", ping_time_bin="20s")
    monkeypatch.setenv("EPA_DEPTH_WITH_MVBS", "1")
    with _lib.launch_trace() as tr:
        ds = ep.calibrate.compute_Sv(ed)
        ep.consolidate.add_depth(ds, depth_offset=3.0, tilt=10.0)
        mv = ep.commongrid.compute_MVBS(ds, range_var="depth", range_bin="2m", ping_time_bin="20s")
        mv["Sv"].values
    assert _sample_kernels(tr) == ["fused_sv_mvbs_kernel"], tr.kernels
    assert ds["depth"].data.materialized
    np.testing.assert_array_equal(ds["depth"].values, ref["depth"].values)
    np.testing.assert_array_equal(ds["Sv"].values, ref["Sv"].values)
    np.testing.assert_allclose(mv["Sv"].values, mv0["Sv"].values, rtol=1e-12, atol=1e-12)
    assert ds["depth"].data.cached_stats() == ref["depth"].data.cached_stats()


@pytest.mark.parametrize("depth", [False, True])
def test_a_ping_with_a_nan_coefficient_is_skipped_not_added(ep, depth):
    """Finite raw samples under a NaN gain: Sv of that ping is NaN, the nanmean skips it (commongrid/utils.py:614-627) --
    a time bin that also holds valid pings keeps the mean of those, a bin of such pings only is empty.  (The fused
    kernel's lean path must not take such a ping: ADVICE round 5.)"""
    import torch
    from echopype_amd import _lib, ops

    d = _case(ep, C=2, P=120, S=1000, ss_every=1000)
    d["backscatter_r"][:] = np.where(np.isnan(d["backscatter_r"]), np.float32(-70.0), d["backscatter_r"])  # every sample finite
    ed = ep.echodata.from_ek60_arrays(d)
    cal = ep.calibrate.api.CALIBRATOR["EK60"](ed, None, None, None, dtype="float64")
    raw, coef, flags, _ = cal._power_inputs("Sv")
    coef = coef.clone()
    coef[0, 7, _lib.CF_G] = float("nan")          # one ping inside a bin of valid pings
    coef[1, 40:60, _lib.CF_A0] = float("inf")     # a whole time bin (20 pings of 1 s)
    C, P, S = raw.shape
    ns = d["ping_time"].astype("datetime64[ns]").astype(np.int64)
    e0, dt = int(ns[0]), 20 * 10**9
    n_t = int((ns[-1] - e0) // dt) + 1
    bs = ops.time_bin_offsets(ops.to_device(ns), e0, dt, n_t)
    one = torch.ones((C, P), dtype=torch.float64, device="cuda")
    zero = torch.zeros((C, P), dtype=torch.float64, device="cuda")
    with _lib.launch_trace() as tr:
        if depth:
            res = ops.sv_mvbs_fused_depth(raw, coef, one, zero, bs, n_t, 5.0, 60)
        else:
            res = ops.sv_mvbs_fused(raw, coef, bs, n_t, 5.0, 60, want_range_stats=True)
    assert "fused_sv_mvbs_kernel" in tr.kernels
    sv = res["Sv"].cpu().numpy()
    assert np.isnan(sv[0, 7, 3:]).all() and np.isinf(sv[1, 40:60, 3:]).all() and np.isfinite(sv[0, 8, 3:]).all()
    # the generic kernel on the same Sv array and the echo_range: the reference's skipna mean
    rng = ops.range_power(raw, coef)
    exp = ops.mvbs(res["Sv"], bs, n_t, 5.0, 60, range=rng)["MVBS"].cpu().numpy()
    got = res["MVBS"].cpu().numpy()
    np.testing.assert_array_equal(np.isnan(got), np.isnan(exp))
    fin = np.isfinite(exp)
    np.testing.assert_allclose(got[fin], exp[fin], rtol=1e-12)
    np.testing.assert_array_equal(np.isinf(got), np.isinf(exp))
    assert np.isfinite(got[0, 0]).any()
