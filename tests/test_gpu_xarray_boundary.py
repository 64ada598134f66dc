"""xarray in -> xarray out (the drop-in keeps the reference's Dataset-in / Dataset-out signatures).  Every
test runs twice: with tests/fake_xarray.py, a duck-typed stand-in for the public API subset the boundary touches (xarray
cannot be installed in the build image: no network), and with the REAL ``xarray`` -- Datasets in an ``xarray.DataTree``
behind an EchoData-like wrapper -- wherever it is importable (``pytest.importorskip``): any box that has xarray proves
the claim.  The library under test is patched in as ``echopype_amd.xr_lite._xr``: an "xarray" EchoData / Dataset goes
in, "xarray" Datasets come out, and remove_background_noise writes Sv_noise / Sv_corrected into the CALLER's dataset as
the reference does (/root/reference/echopype/clean/api.py:490-502)."""
import numpy as np
import pytest

import fake_xarray

pytestmark = pytest.mark.gpu
DIMS = ("channel", "ping_time", "range_sample")
fx = fake_xarray  # rebound by the ``ep`` fixture to the library of the running parametrisation


@pytest.fixture(params=["stand-in", "xarray"])
def ep(request, monkeypatch):
    import torch

    if not torch.cuda.is_available():
        pytest.fail("these tests need a GPU")
    import converter_layout
    import echopype_amd
    from echopype_amd import xr_lite

    lib = fake_xarray if request.param == "stand-in" else pytest.importorskip("xarray")
    monkeypatch.setattr(xr_lite, "_xr", lib)
    monkeypatch.setitem(globals(), "fx", lib)
    prev = converter_layout.use(lib)
    yield echopype_amd
    converter_layout.use(prev)


def _as_fake(ds):
    """lite Dataset -> the stand-in (what a user holding real xarray objects would pass)."""
    return fx.Dataset({k: (v.dims, np.asarray(v.values), dict(v.attrs)) for k, v in ds.data_vars.items()},
                      coords={k: (c.dims, np.asarray(c.values), dict(c.attrs)) for k, c in ds.coords.items()},
                      attrs=dict(ds.attrs))


class ForeignEchoData:
    """Looks like echopype's EchoData: sonar_model + group access returning "xarray" Datasets."""

    def __init__(self, lite_ed):
        self.sonar_model, self.source_file, self.converted_raw_path = lite_ed.sonar_model, lite_ed.source_file, None
        self._g = {g: _as_fake(lite_ed[g]) for g in lite_ed.group_paths}
        self.group_paths = list(self._g)  # (Datasets of the library under test: the stand-in's or xarray's)

    def __getitem__(self, g):
        return self._g[g]


def test_xarray_in_xarray_out_through_the_chain(ep):
    d = ep.synth.ek60_numpy(2, 60, 300)
    lite_ed = ep.echodata.from_ek60_arrays(d)
    ref_sv = ep.calibrate.compute_Sv(lite_ed)                      # lite in -> lite out, device resident
    assert isinstance(ref_sv, ep.Dataset) and ep.xr_lite.is_device(ref_sv["Sv"].data)
    ds = ep.calibrate.compute_Sv(ForeignEchoData(lite_ed))         # "xarray" EchoData in -> "xarray" Dataset out
    assert isinstance(ds, fx.Dataset) and isinstance(ds["Sv"].values, np.ndarray)
    np.testing.assert_array_equal(ds["Sv"].values, ref_sv["Sv"].values)
    np.testing.assert_array_equal(ds["echo_range"].values, ref_sv["echo_range"].values)
    assert ds["Sv"].dims == DIMS and set(ds.coords) >= set(DIMS)
    # remove_background_noise: adds to the caller's dataset and returns it with the provenance attributes
    before = set(ds.data_vars)
    out = ep.clean.remove_background_noise(ds, 20, 50)
    assert {"Sv_noise", "Sv_corrected"} <= set(ds.data_vars) - before          # the CALLER's object was extended
    assert isinstance(out, fx.Dataset) and out.attrs["processing_function"] == "clean.remove_background_noise"
    ref = ep.clean.remove_background_noise(ref_sv, 20, 50)
    # (the block means are accumulated with floating-point atomics: two runs agree to rounding, not bit for bit)
    np.testing.assert_allclose(ds["Sv_corrected"].values, ref["Sv_corrected"].values, rtol=1e-12, equal_nan=True)
    assert ds["Sv_noise"].attrs == ref["Sv_noise"].attrs
    # compute_MVBS / estimate_background_noise / masks / apply_mask: "xarray" in -> "xarray" out
    mv = ep.commongrid.compute_MVBS(ds, range_bin="2m", ping_time_bin="20s")
    assert isinstance(mv, fx.Dataset)
    np.testing.assert_allclose(mv["Sv"].values, ep.commongrid.compute_MVBS(ref_sv, range_bin="2m", ping_time_bin="20s")["Sv"].values,
                               rtol=1e-12, equal_nan=True)
    sn = ep.clean.estimate_background_noise(ds, 20, 50)
    assert isinstance(sn, fx.DataArray) and sn.dims == DIMS
    m = ep.clean.mask_impulse_noise(ds, range_var="echo_range", use_index_binning=True)
    assert isinstance(m, fx.DataArray) and m.values.dtype == bool
    masked = ep.mask.apply_mask(ds, m)
    assert isinstance(masked, fx.Dataset)
    # the fused entry point returns a pair
    a, b = ep.compute_Sv_MVBS(ForeignEchoData(lite_ed), range_bin="2m", ping_time_bin="20s")
    assert isinstance(a, fx.Dataset) and isinstance(b, fx.Dataset)
    np.testing.assert_allclose(b["Sv"].values, mv["Sv"].values, rtol=1e-12, equal_nan=True)


def test_first_argument_by_keyword_and_add_depth_in_place(ep):
    """The reference's functions take their dataset by keyword too (compute_Sv(echodata=ed), compute_MVBS(ds_Sv=ds),
    apply_mask(source_ds=..., mask=...)); add_depth assigns ds["depth"] on the CALLER's dataset
    (/root/reference/echopype/consolidate/api.py:221-241) and converts a foreign ``echodata=``."""
    d = ep.synth.ek60_numpy(2, 40, 200)
    lite_ed = ep.echodata.from_ek60_arrays(d)
    fed = ForeignEchoData(lite_ed)
    ds = ep.calibrate.compute_Sv(echodata=fed)
    assert isinstance(ds, fx.Dataset)
    ref = ep.calibrate.compute_Sv(echodata=lite_ed)
    np.testing.assert_array_equal(ds["Sv"].values, ref["Sv"].values)
    mv = ep.commongrid.compute_MVBS(ds_Sv=ds, range_bin="2m", ping_time_bin="20s")
    assert isinstance(mv, fx.Dataset)
    np.testing.assert_allclose(mv["Sv"].values, ep.commongrid.compute_MVBS(ds_Sv=ref, range_bin="2m", ping_time_bin="20s")["Sv"].values,
                               rtol=1e-12, equal_nan=True)
    m = ep.clean.mask_impulse_noise(ds_Sv=ds, range_var="echo_range", use_index_binning=True)
    assert isinstance(ep.mask.apply_mask(source_ds=ds, mask=m), fx.Dataset)
    with pytest.raises(TypeError, match="missing 1 required positional argument"):
        ep.commongrid.compute_MVBS(range_bin="2m")
    out = ep.consolidate.add_depth(ds=ds, echodata=fed, depth_offset=3.5)
    assert "depth" in ds.data_vars and isinstance(out, fx.Dataset)           # the caller's dataset carries depth now
    np.testing.assert_allclose(ds["depth"].values, ds["echo_range"].values + 3.5, rtol=1e-15, equal_nan=True)
    assert "history" in ds["depth"].attrs


# ---- EchoData as the converter writes it: a DataTree-like container, every group, the converter's dtypes ---------------
def test_converter_shaped_echodata_through_the_chain(ep):
    """tests/converter_layout.py: EK60 (float32 power + angle planes, byte flags, Environment on (channel, time1)), EK80 BB
    and CW (float64 r / i planes with a string-labelled beam dimension, filter tables on (channel, filter_time, n), one
    Environment timestamp), AZFP -- through compute_Sv / compute_TS, remove_background_noise and compute_MVBS: "xarray"
    out, values identical to the lite-container path on the same numbers."""
    import converter_layout as cl
    from test_gpu_api import _ek80

    # EK60
    lite = ep.echodata.from_ek60_arrays(ep.synth.ek60_numpy(3, 50, 400, vary_tau=True))
    ed = cl.ek60(lite)
    assert ed["Sonar/Beam_group2"] is None and "Sonar/Beam_group1" in ed.group_paths and ed["Top-level"] is not None
    ref = ep.calibrate.compute_Sv(lite)
    ds = ep.calibrate.compute_Sv(ed)
    assert isinstance(ds, fx.Dataset)
    np.testing.assert_array_equal(ds["Sv"].values, ref["Sv"].values)
    np.testing.assert_array_equal(ds["echo_range"].values, ref["echo_range"].values)
    np.testing.assert_array_equal(ep.calibrate.compute_TS(ed)["TS"].values, ep.calibrate.compute_TS(lite)["TS"].values)
    out = ep.clean.remove_background_noise(ds, 10, 40)
    ref2 = ep.clean.remove_background_noise(ref, 10, 40)
    np.testing.assert_allclose(out["Sv_corrected"].values, ref2["Sv_corrected"].values, rtol=1e-12, equal_nan=True)
    mv = ep.commongrid.compute_MVBS(ds, range_bin="5m", ping_time_bin="10s")
    np.testing.assert_allclose(mv["Sv"].values, ep.commongrid.compute_MVBS(ref, range_bin="5m", ping_time_bin="10s")["Sv"].values,
                               rtol=1e-12, equal_nan=True)
    # EK80 complex, BB and CW
    for wf in ("BB", "CW"):
        d, filt = _ek80(ep, wf, C=2, P=9, S=700, mixed_nan=True)
        lite = ep.echodata.from_ek80_arrays(d, filt, filter_time_idx=[0])
        ed = cl.ek80(lite)
        kw = dict(waveform_mode=wf, encode_mode="complex")
        ref = ep.calibrate.compute_Sv(lite, **kw)
        ds = ep.calibrate.compute_Sv(ed, **kw)
        assert isinstance(ds, fx.Dataset) and ds["Sv"].dims == DIMS
        np.testing.assert_array_equal(ds["Sv"].values, ref["Sv"].values)
        np.testing.assert_array_equal(ds["echo_range"].values, ref["echo_range"].values)
        mv = ep.commongrid.compute_MVBS(ds, range_bin="1m", ping_time_bin="5s")
        np.testing.assert_allclose(mv["Sv"].values,
                                   ep.commongrid.compute_MVBS(ref, range_bin="1m", ping_time_bin="5s")["Sv"].values,
                                   rtol=1e-12, equal_nan=True)
    # AZFP (salinity / pressure from the user, as the reference requires)
    lite = ep.echodata.from_azfp_arrays(ep.synth.azfp_numpy(3, 30, 200))
    ed = cl.azfp(lite)
    env = {"salinity": 29.6, "pressure": 60.0}
    ref = ep.calibrate.compute_Sv(lite, env_params=env)
    ds = ep.calibrate.compute_Sv(ed, env_params=env)
    np.testing.assert_array_equal(ds["Sv"].values, ref["Sv"].values)
    out = ep.clean.remove_background_noise(ds, 5, 30)
    assert "Sv_corrected" in ds.data_vars and isinstance(out, fx.Dataset)
