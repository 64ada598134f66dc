"""compute_MVBS and the NASC array pass on a real MI355X against the reference's OWN brute-force expectations,
executed from /root/reference (tests/golden/ref_mvbs_goldens.npz, made by oracle/gen_mvbs_goldens.py from
mock_data.py::_get_expected_mvbs_val and tests/commongrid/conftest.py::_get_expected_nasc_val_nanmean).
Reads like tests/commongrid/test_commongrid_api.py:371-436 and :447-470 (atol = rtol = 1e-10)."""
import numpy as np
import pytest

import mvbs_goldens

pytestmark = pytest.mark.gpu
DIMS = ("channel", "ping_time", "range_sample")


@pytest.fixture(scope="module")
def ep():
    import torch

    if not torch.cuda.is_available():
        pytest.fail("these tests need a GPU")
    import echopype_amd

    return echopype_amd


def _ds(ep, d, dtype="float64", device=False):
    import torch

    C, P, S = d["Sv"].shape
    ds = ep.Dataset(coords={"channel": [f"ch_{i}" for i in range(C)], "ping_time": d["ping_time"],
                            "range_sample": np.arange(S)})
    for k in ("Sv", "echo_range", "depth"):
        a = d[k].astype(dtype)
        ds[k] = (DIMS, ep.DeviceArray(torch.from_numpy(a).cuda()) if device else a)
    ds["frequency_nominal"] = (("channel",), np.arange(C, dtype=np.float64))
    return ds


@pytest.mark.parametrize("device", [False, True])
@pytest.mark.parametrize("tag", mvbs_goldens.CASES)
def test_compute_MVBS_matches_reference_brute_force(ep, tag, device):
    d = mvbs_goldens.load(tag)
    ds = _ds(ep, d, device=device)
    mv = ep.commongrid.compute_MVBS(ds, range_bin=f"{d['range_bin']}m", ping_time_bin=d["ping_time_bin"],
                                    skipna=d["value_nans"] == 0)
    mvbs_goldens.compare_mvbs(np.asarray(mv["Sv"].values), d["mvbs"])
    np.testing.assert_array_equal(np.asarray(mv["ping_time"].values).astype("datetime64[ns]"), d["mvbs_time_labels"])


@pytest.mark.parametrize("tag", ["regular", "irregular"])
def test_compute_MVBS_float32_and_on_depth(ep, tag):
    """float32 Sv within 1e-3; range_var="depth" = the same values one offset deeper."""
    d = mvbs_goldens.load(tag)
    mv = ep.commongrid.compute_MVBS(_ds(ep, d, "float32"), range_bin=f"{d['range_bin']}m", ping_time_bin=d["ping_time_bin"])
    got = np.asarray(mv["Sv"].values, np.float64)
    n = min(got.shape[2], d["mvbs"].shape[2])
    np.testing.assert_array_equal(np.isnan(got[..., :n]), np.isnan(d["mvbs"][..., :n]))
    ok = ~np.isnan(d["mvbs"][..., :n])
    assert np.max(np.abs(got[..., :n][ok] - d["mvbs"][..., :n][ok]) / np.maximum(np.abs(d["mvbs"][..., :n][ok]), 1.0)) < 1e-3
    if tag == "regular" and d["depth_offset"] % d["range_bin"] == 0:  # bins shift by a whole number
        k = int(d["depth_offset"] / d["range_bin"])
        mvd = ep.commongrid.compute_MVBS(_ds(ep, d), range_var="depth", range_bin=f"{d['range_bin']}m",
                                         ping_time_bin=d["ping_time_bin"])
        gd = np.asarray(mvd["Sv"].values)
        assert np.isnan(gd[..., :k]).all()
        mvbs_goldens.compare_mvbs(gd[..., k:], d["mvbs"])


@pytest.mark.parametrize("dtype,tol", [("float64", 1e-10), ("float32", 1e-3)])
@pytest.mark.parametrize("tag", mvbs_goldens.CASES)
def test_nasc_kernel_matches_reference_brute_force(ep, tag, dtype, tol):
    import torch

    d = mvbs_goldens.load(tag)
    r_edges = np.arange(0, np.nanmax(d["depth"]) + d["nasc_range_bin"], d["nasc_range_bin"])
    d_edges = np.arange(0, d["distance_nmi"].max() + d["dist_bin"], d["dist_bin"])
    starts = np.searchsorted(d["distance_nmi"], d_edges, side="left").astype(np.int32)
    got = ep.ops.nasc(torch.from_numpy(d["Sv"].astype(dtype)).cuda(), torch.from_numpy(d["depth"].astype(dtype)).cuda(),
                      torch.from_numpy(starts).cuda(), len(d_edges) - 1, d["nasc_range_bin"], len(r_edges) - 1)
    got = (got[0] if isinstance(got, tuple) else got).cpu().numpy().astype(np.float64)
    assert got.shape == d["nasc"].shape
    np.testing.assert_array_equal(np.isnan(got), np.isnan(d["nasc"]))
    np.testing.assert_allclose(got, d["nasc"], rtol=tol, atol=1e-10 if dtype == "float64" else 1e-3 * np.nanmax(d["nasc"]),
                               equal_nan=True)
