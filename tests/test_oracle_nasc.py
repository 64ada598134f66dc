"""Oracle compute_NASC (oracle/nasc.py) pinned by closed forms (geodesic) and by the reference's
known-answer tests restated (tests/commongrid/test_commongrid_api.py:155-167 and the brute-force loops
of tests/commongrid/conftest.py:466-546)."""
import numpy as np
import pytest
import scipy.integrate

from oracle import nasc as onasc


def test_geodesic_closed_forms():
    a, f = onasc.WGS84_A, onasc.WGS84_F
    e2 = f * (2 - f)
    # along the equator the geodesic is the equator itself (for arcs well below the antipodal limit)
    for dlon in (1e-5, 0.01, 1.0, 30.0):
        np.testing.assert_allclose(onasc.geodesic_m(0.0, 10.0, 0.0, 10.0 + dlon), a * np.radians(dlon), rtol=1e-12, atol=1e-6)
    # along a meridian: quadrature of the meridional radius of curvature M(phi)
    M = lambda p: a * (1 - e2) / (1 - e2 * np.sin(p) ** 2) ** 1.5  # noqa: E731
    for lat1, lat2 in ((0.0, 1.0), (42.48916859, 42.49071833), (-33.0, 12.5), (60.0, 89.0)):
        arc, _ = scipy.integrate.quad(M, np.radians(lat1), np.radians(lat2), epsabs=1e-6, epsrel=1e-14)
        np.testing.assert_allclose(onasc.geodesic_m(lat1, -124.9, lat2, -124.9), arc, rtol=1e-11, atol=1e-6)
    np.testing.assert_allclose(onasc.geodesic_m(0.0, 0.0, 90.0, 0.0), 10001965.729, atol=1e-3)  # quarter meridian
    # symmetry, identity, and a short oblique line against the local flat-earth metric
    p, q = (42.48916859, -124.88296688), (42.49071833, -124.81919229)
    assert onasc.geodesic_m(*p, *q) == pytest.approx(onasc.geodesic_m(*q, *p), rel=1e-13)
    assert onasc.geodesic_m(*p, *p) == 0.0
    lat = np.radians(42.49)
    N = a / np.sqrt(1 - e2 * np.sin(lat) ** 2)
    dN, dE = M(lat) * np.radians(1e-4), N * np.cos(lat) * np.radians(1e-4)
    np.testing.assert_allclose(onasc.geodesic_m(42.49 - 5e-5, -124.8 - 5e-5, 42.49 + 5e-5, -124.8 + 5e-5),
                               np.hypot(dN, dE), rtol=1e-9)


def test_distance_from_latlon_semantics():
    lat = np.array([10.0, 10.001, np.nan, 10.003, 10.004, 10.004])
    lon = np.full(6, 20.0)
    d = onasc.distance_from_latlon(lat, lon)
    step = onasc.geodesic_m(10.0, 20.0, 10.001, 20.0) / 1852
    s34 = onasc.geodesic_m(10.003, 20.0, 10.004, 20.0) / 1852
    # row 0: d(0,1); rows 1, 2 dropped (NaN on one side) -> forward filled; row 3: + d(3,4); row 4: + 0;
    # row 5 has no successor -> forward filled
    np.testing.assert_allclose(d, [step, step, step, step + s34, step + s34, step + s34], rtol=1e-12)
    with pytest.raises(ValueError, match="All lat/lon entries are NaN!"):
        onasc.distance_from_latlon(np.full(3, np.nan), np.zeros(3))


def test_simple_NASC_echoview_value():
    """tests/commongrid/test_commongrid_api.py:155-167 with conftest.py:404-464."""
    rng = np.random.default_rng(42)
    r = np.array([0.5, 1.5, 2.5, 3.5, 9])
    sv0 = np.array([[1.0, 2.0, 3.0, 4.0, np.nan], [6.0, 7.0, 8.0, 9.0, 10.0], [11.0, 12.0, 13.0, 14.0, 15.0],
                    [16.0, 17.0, 18.0, 19.0, np.nan], [21.0, 22.0, 23.0, 24.0, 25.0]])  # (range_sample, distance)
    for _ in range(2):
        sv = sv0 + rng.random() * 5
        Sv = 10 * np.log10(sv).T[None]  # (1, P=5 distances, S=5)
        depth = np.tile(r, (1, 5, 1))
        t = np.datetime64("2020-01-01") + np.arange(5) * np.timedelta64(1, "m")
        nasc, _ = onasc.compute_raw_NASC(Sv, depth, np.arange(5.0), t, np.array([1, 5]), np.array([-5, 10]))
        r0, r1 = np.argmin(abs(r - 2)), np.argmin(abs(r - 20))
        sh = np.r_[np.diff(r), np.nan]
        echoview = np.nanmean(sv[r0:r1]) * np.sum(sh[r0:r1]) * 4 * np.pi * 1852**2
        np.testing.assert_allclose(nasc[0, 0, 0], echoview, rtol=1e-10, atol=1e-10)


@pytest.mark.parametrize("irregular", [False, True])
@pytest.mark.parametrize("closed", ["left", "right"])
def test_compute_NASC_against_brute_force(irregular, closed):
    """conftest.py:466-546 (_get_expected_nasc_val_nanmean) restated with explicit loops."""
    rng = np.random.default_rng(7)
    C, P, S = 2, 40, 30
    lat = np.linspace(42.48916859, 42.49071833, P)  # conftest.py:106-107
    lon = np.linspace(-124.88296688, -124.81919229, P)
    depth = np.tile(np.arange(S) * 0.5 + 0.25, (C, P, 1))
    Sv = 10 * np.log10(rng.random((C, P, S)) + 0.1)
    if irregular:
        depth = depth + 0.1 * rng.random((C, P, S)).cumsum(axis=2)
        depth[1, :, -4:] = np.nan
        Sv[rng.random((C, P, S)) < 0.1] = np.nan
    t = np.datetime64("2020-01-01T00:00:00", "ns") + np.arange(P) * np.timedelta64(1, "s")
    out = onasc.compute_NASC(Sv, depth, lat, lon, t, range_bin=2.0, dist_bin=0.5, closed=closed)
    dist = out["distance_nmi"]
    d_edges = np.arange(0, dist.max() + 0.5, 0.5)
    r_edges = np.arange(0, np.nanmax(depth) + 2.0, 2.0)
    assert out["NASC"].shape == (C, len(d_edges) - 1, len(r_edges) - 1)
    assert len(d_edges) - 1 == int(np.ceil(dist.max() / 0.5))  # test_commongrid_api.py:151
    inside = (lambda x, lo, hi: (x >= lo) & (x < hi)) if closed == "left" else (lambda x, lo, hi: (x > lo) & (x <= hi))
    sv = 10 ** (Sv / 10)
    for c in range(C):
        for i in range(len(d_edges) - 1):
            pings = inside(dist, d_edges[i], d_edges[i + 1])
            for j in range(len(r_edges) - 1):
                cell = pings[:, None] & inside(depth[c], r_edges[j], r_edges[j + 1])
                vals = sv[c][cell]
                vals = vals[~np.isnan(vals)]
                got = out["NASC"][c, i, j]
                if vals.size == 0:
                    assert np.isnan(got)
                    continue
                dd = np.diff(depth[c], axis=1)
                h = np.nansum(dd[cell[:, :-1]]) / pings.sum()
                np.testing.assert_allclose(got, vals.mean() * h * 4 * np.pi * 1852**2, rtol=1e-10, atol=1e-10)
    # per-bin mean positions and ping_time
    for i in range(len(d_edges) - 1):
        pings = inside(dist, d_edges[i], d_edges[i + 1])
        if pings.any():
            np.testing.assert_allclose(out["latitude"][i], lat[pings].mean(), rtol=1e-14)
            np.testing.assert_allclose(out["ping_time"][i], t[pings].astype(np.int64).mean(), rtol=1e-15)


def test_skipna_false_poisons_cells():
    rng = np.random.default_rng(3)
    Sv = 10 * np.log10(rng.random((1, 6, 8)) + 0.1)
    Sv[0, 2, 3] = np.nan
    depth = np.tile(np.arange(8.0) + 0.5, (1, 6, 1))
    t = np.datetime64("2020-01-01", "ns") + np.arange(6) * np.timedelta64(1, "s")
    a, _ = onasc.compute_raw_NASC(Sv, depth, np.zeros(6), t, np.arange(0, 10, 2.0), np.array([0.0, 0.5]), skipna=True)
    b, _ = onasc.compute_raw_NASC(Sv, depth, np.zeros(6), t, np.arange(0, 10, 2.0), np.array([0.0, 0.5]), skipna=False)
    assert np.isnan(b[0, 0, 1]) and np.isfinite(a[0, 0, 1])
    np.testing.assert_array_equal(np.delete(a, 1, axis=2), np.delete(b, 1, axis=2))
