"""compute_NASC on a real MI355X: the epa_nasc kernel through the C ABI and the
drop-in commongrid.compute_NASC vs the oracle (oracle/nasc.py); reads like the reference's
tests/commongrid/test_commongrid_api.py:97-167,447-470."""
import numpy as np
import pytest

from oracle import nasc as onasc

pytestmark = pytest.mark.gpu
DIMS = ("channel", "ping_time", "range_sample")


@pytest.fixture(scope="module")
def ep():
    import torch

    if not torch.cuda.is_available():
        pytest.fail("these tests need a GPU")
    import echopype_amd

    return echopype_amd


def _survey(C, P, S, seed, irregular=True):
    rng = np.random.default_rng(seed)
    lat = np.linspace(42.48916859, 42.52071833, P)  # tests/commongrid/conftest.py:106-107, longer track
    lon = np.linspace(-124.88296688, -124.81919229, P)
    depth = np.tile(np.arange(S) * 0.5 + 0.25, (C, P, 1))
    Sv = 10 * np.log10(rng.random((C, P, S)) * 1e-6 + 1e-8)
    if irregular:
        depth = depth + 0.1 * rng.random((C, P, S)).cumsum(axis=2)
        depth[-1, :, -S // 8:] = np.nan
        Sv[rng.random((C, P, S)) < 0.1] = np.nan
        lat[P // 3] = np.nan  # a position gap: the pair before and the pair after are dropped
    t = np.datetime64("2020-01-01T00:00:00", "ns") + np.arange(P) * np.timedelta64(1, "s")
    return Sv, depth, lat, lon, t


def _close(got, exp, rtol):
    got, exp = np.asarray(got, np.float64), np.asarray(exp, np.float64)
    assert got.shape == exp.shape
    np.testing.assert_array_equal(np.isnan(got), np.isnan(exp))
    ok = ~np.isnan(exp)
    np.testing.assert_allclose(got[ok], exp[ok], rtol=rtol, atol=0)


@pytest.mark.parametrize("dtype,rtol", [("float64", 1e-10), ("float32", 1e-3)])
@pytest.mark.parametrize("closed", ["left", "right"])
@pytest.mark.parametrize("skipna", [True, False])
def test_nasc_kernel_vs_oracle(ep, dtype, rtol, closed, skipna):
    import torch

    Sv, depth, lat, lon, t = _survey(3, 300, 160, 1)
    Sv, depth = Sv.astype(dtype), depth.astype(dtype)
    dist = np.sort(np.random.default_rng(2).random(300) * 3.0)
    dist[:4] = 0.0
    r_edges = np.arange(0, np.nanmax(depth.astype(np.float64)) + 5.0, 5.0)
    d_edges = np.arange(0, dist.max() + 0.5, 0.5)
    exp, _ = onasc.compute_raw_NASC(Sv.astype(np.float64), depth.astype(np.float64), dist, t, r_edges, d_edges,
                                    skipna=skipna, closed=closed)
    starts = np.searchsorted(dist, d_edges, side="left" if closed == "left" else "right").astype(np.int32)
    got, svm, hm = ep.ops.nasc(torch.from_numpy(Sv).cuda(), torch.from_numpy(depth).cuda(),
                               torch.from_numpy(starts).cuda(), len(d_edges) - 1, 5.0, len(r_edges) - 1,
                               skipna=skipna, closed=closed, want_parts=True)
    _close(got.cpu().numpy(), exp, rtol)
    np.testing.assert_allclose(np.nan_to_num(svm.cpu().numpy() * hm.cpu().numpy() * 4 * np.pi * 1852**2),
                               np.nan_to_num(exp), rtol=max(rtol, 1e-6))


@pytest.mark.parametrize("irregular", [False, True])
def test_compute_NASC_values(ep, irregular):
    C, P, S = 2, 400, 120
    Sv, depth, lat, lon, t = _survey(C, P, S, 3, irregular)
    ds = ep.Dataset(coords={"channel": ["a", "b"], "ping_time": t, "range_sample": np.arange(S)})
    ds["Sv"], ds["depth"] = (DIMS, Sv), (DIMS, depth)
    ds["latitude"] = (("ping_time",), lat, {"standard_name": "latitude"})
    ds["longitude"] = (("ping_time",), lon, {"standard_name": "longitude"})
    ds["frequency_nominal"] = (("channel",), np.array([38e3, 120e3]))
    out = ep.commongrid.compute_NASC(ds, range_bin="2m", dist_bin="0.5nmi")
    exp = onasc.compute_NASC(Sv, depth, lat, lon, t, 2.0, 0.5)
    assert out["NASC"].dims == ("channel", "distance", "depth")
    np.testing.assert_array_equal(out["channel"].values, ds["channel"].values)
    assert out["depth"].values.size == np.ceil(np.nanmax(depth) / 2.0)           # test_commongrid_api.py:150
    assert out["distance"].values.size == np.ceil(exp["distance_nmi"].max() / 0.5)  # :151
    _close(out["NASC"].values, exp["NASC"], 1e-10)
    np.testing.assert_array_equal(out["distance"].values, exp["distance"])
    np.testing.assert_array_equal(out["depth"].values, exp["depth"])
    np.testing.assert_allclose(out["latitude"].values, exp["latitude"], rtol=1e-14)
    np.testing.assert_allclose(out["longitude"].values, exp["longitude"], rtol=1e-14)
    np.testing.assert_allclose(out["ping_time"].values.astype(np.int64), exp["ping_time"], rtol=0, atol=1024)  # the oracle averages float64 nanoseconds (ulp 256 ns)
    assert out["NASC"].attrs["units"] == "m2 nmi-2" and out.attrs["Conventions"] == "CF-1.7,ACDD-1.3"
    assert out.attrs["geospatial_lat_min"] == round(float(np.nanmin(lat)), 5)
    assert out["latitude"].attrs["standard_name"] == "latitude"
    assert np.isfinite(out["NASC"].values).any()


def test_compute_NASC_from_MVBS_dataset_and_errors(ep):
    """test_commongrid_api.py:97-151 with compute_mvbs=True: NASC of a dataset already gridded on depth."""
    C, P, S = 2, 600, 200
    Sv, depth, lat, lon, t = _survey(C, P, S, 5, irregular=False)
    ds = ep.Dataset(coords={"channel": ["a", "b"], "ping_time": t, "range_sample": np.arange(S)})
    ds["Sv"], ds["depth"] = (DIMS, Sv), (DIMS, depth)
    ds["latitude"], ds["longitude"] = (("ping_time",), lat), (("ping_time",), lon)
    mvbs = ep.commongrid.compute_MVBS(ds, range_var="depth", range_bin="2m", ping_time_bin="5s")
    assert "latitude" in mvbs and mvbs["Sv"].dims == ("channel", "ping_time", "depth")
    out = ep.commongrid.compute_NASC(mvbs, range_bin="10m", dist_bin="0.5nmi")
    m = mvbs["Sv"].values
    d3 = np.broadcast_to(mvbs["depth"].values[None, None, :], m.shape)
    exp = onasc.compute_NASC(m, d3, mvbs["latitude"].values, mvbs["longitude"].values, mvbs["ping_time"].values,
                             10.0, 0.5)
    _close(out["NASC"].values, exp["NASC"], 1e-10)

    with pytest.raises(ValueError, match="Input Sv dataset must contain all of the following variables"):
        ep.commongrid.compute_NASC(ds.drop_vars("latitude"))
    with pytest.raises(TypeError, match="dist_bin must be a string"):
        ep.commongrid.compute_NASC(ds, dist_bin=0.5)
    with pytest.raises(ValueError, match="Distance bin must be in nautical miles"):
        ep.commongrid.compute_NASC(ds, dist_bin="0.5km")
    with pytest.raises(TypeError, match="range_bin must be a string"):
        ep.commongrid.compute_NASC(ds, range_bin=10)
    with pytest.raises(ValueError, match="is not a valid option"):
        ep.commongrid.compute_NASC(ds, closed="both")
    bad = ds.copy()
    bad["latitude"] = (("ping_time",), np.full(P, np.nan))
    with pytest.raises(ValueError, match="All lat/lon entries are NaN!"):
        ep.commongrid.compute_NASC(bad)


def test_nasc_depth_grid_beyond_the_lds_accumulators(ep):
    """7 000+ depth bins (0.1 m over 700 m) do not fit the per-workgroup LDS cells: the kernel accumulates straight
    into the global workspace with atomics -- same answer."""
    import torch

    rng = np.random.default_rng(5)
    C, P, S = 2, 40, 3000
    depth = np.tile(np.arange(S) * 0.2347 + 0.1, (C, P, 1))
    Sv = 10 * np.log10(rng.random((C, P, S)) * 1e-6 + 1e-8)
    Sv[rng.random((C, P, S)) < 0.05] = np.nan
    dist = np.sort(rng.random(P) * 2.0)
    t = np.datetime64("2020-01-01T00:00:00", "ns") + np.arange(P) * np.timedelta64(1, "s")
    r_edges = np.arange(0, depth.max() + 0.1, 0.1)
    d_edges = np.arange(0, dist.max() + 0.5, 0.5)
    assert len(r_edges) - 1 > 6500
    exp, _ = onasc.compute_raw_NASC(Sv, depth, dist, t, r_edges, d_edges)
    starts = np.searchsorted(dist, d_edges, side="left").astype(np.int32)
    got = ep.ops.nasc(torch.from_numpy(Sv).cuda(), torch.from_numpy(depth).cuda(), torch.from_numpy(starts).cuda(),
                      len(d_edges) - 1, 0.1, len(r_edges) - 1)
    _close(got.cpu().numpy(), exp, 1e-10)


def test_geodesic_steps_kernel_equals_the_host_vincenty(ep):
    """epa_geodesic_steps (one launch) against commongrid.utils.geodesic_distance_m (vectorised NumPy, itself held to
    geopy's numbers in tests/test_host_logic.py): a track with long and short steps, a pole-ward leg, coincident points,
    NaN positions; and get_distance_from_latlon gives the same cumulative distance on either side of the size switch."""
    import torch

    from echopype_amd import ops
    from echopype_amd.commongrid import utils as cgu

    rng = np.random.default_rng(5)
    P = 20_000
    lat = 45.0 + np.cumsum(rng.normal(0, 2e-5, P))
    lon = -125.0 + np.cumsum(rng.normal(1e-5, 3e-5, P))
    lat[1000:1010] = lat[1000]                # coincident points
    lon[1000:1010] = lon[1000]
    lat[5000:5300] += np.linspace(0, 35, 300)  # a long pole-ward leg (large steps)
    lat[7000], lon[7013] = np.nan, np.nan
    got = ops.geodesic_steps(torch.from_numpy(lat).cuda(), torch.from_numpy(lon).cuda()).cpu().numpy()
    exp = cgu.geodesic_distance_m(lat[:-1], lon[:-1], lat[1:], lon[1:])
    assert np.isnan(got[-1])
    np.testing.assert_array_equal(np.isnan(got[:-1]), np.isnan(exp))
    f = ~np.isnan(exp)
    assert np.all(got[:-1][f][exp[f] == 0] == 0)
    # (both iterate lambda to 1e-14 rad = 6e-8 m on the ellipsoid; the NumPy form keeps iterating EVERY pair until the
    #  last one has converged, the kernel stops pair by pair: they agree to the iteration's own tolerance)
    np.testing.assert_allclose(got[:-1][f], exp[f], rtol=1e-11, atol=2e-7)
    ds = ep.Dataset(coords={"ping_time": np.arange(P)})
    ds["latitude"] = (("ping_time",), lat)
    ds["longitude"] = (("ping_time",), lon)
    on_device = cgu.get_distance_from_latlon(ds)
    old, cgu._GEODESIC_ON_DEVICE = cgu._GEODESIC_ON_DEVICE, 10**9
    try:
        on_host = cgu.get_distance_from_latlon(ds)
    finally:
        cgu._GEODESIC_ON_DEVICE = old
    np.testing.assert_allclose(on_device, on_host, rtol=1e-12, atol=2e-7 * P / 1852.0)
