"""Oracle: compute_NASC (test infrastructure).

Restates /root/reference/echopype/commongrid/api.py:269-416 (compute_NASC) and
commongrid/utils.py:97-205 (compute_raw_NASC), :208-231 (get_distance_from_latlon).

Third-party arithmetic restated here:
  * geopy.distance.distance (geopy, requirements.txt, un-vendored, absent from this image): the
    geodesic on the WGS-84 ellipsoid (a = 6378137 m, f = 1/298.257223563), in nautical miles
    (metres / 1852).  Restated with Vincenty's inverse formula (Survey Review XXIII, 1975), which
    agrees with geopy's Karney solution to well below a millimetre for the ping-to-ping distances
    of this path.  Pinned by closed forms in tests/test_oracle_nasc.py: arcs of the equator
    (a * dlon), meridian arcs against a quadrature of the meridional radius of curvature, symmetry.
  * flox nanmean / nansum group-bys: as in oracle/commongrid.py.
Parity of the NASC assembly is pinned by the reference's own known-answer tests restated in
tests/test_oracle_nasc.py (tests/commongrid/test_commongrid_api.py:155-167 Echoview value;
tests/commongrid/conftest.py:466-546 brute-force loops).
"""
import math
import warnings

import numpy as np
import pandas as pd

from .commongrid import bin_index

__all__ = ["geodesic_m", "distance_from_latlon", "compute_raw_NASC", "compute_NASC"]

WGS84_A = 6378137.0
WGS84_F = 1 / 298.257223563


def geodesic_m(lat1, lon1, lat2, lon2):
    """Vincenty inverse on WGS-84, scalar, metres."""
    a, f = WGS84_A, WGS84_F
    b = (1 - f) * a
    if lat1 == lat2 and lon1 == lon2:
        return 0.0
    U1 = math.atan((1 - f) * math.tan(math.radians(lat1)))
    U2 = math.atan((1 - f) * math.tan(math.radians(lat2)))
    L = math.radians(lon2 - lon1)
    sU1, cU1, sU2, cU2 = math.sin(U1), math.cos(U1), math.sin(U2), math.cos(U2)
    lam = L
    for _ in range(200):
        sl, cl = math.sin(lam), math.cos(lam)
        sin_sig = math.hypot(cU2 * sl, cU1 * sU2 - sU1 * cU2 * cl)
        if sin_sig == 0:
            return 0.0
        cos_sig = sU1 * sU2 + cU1 * cU2 * cl
        sig = math.atan2(sin_sig, cos_sig)
        sin_al = cU1 * cU2 * sl / sin_sig
        cos2_al = 1 - sin_al**2
        cos_2sm = cos_sig - 2 * sU1 * sU2 / cos2_al if cos2_al != 0 else 0.0
        C = f / 16 * cos2_al * (4 + f * (4 - 3 * cos2_al))
        lam_new = L + (1 - C) * f * sin_al * (
            sig + C * sin_sig * (cos_2sm + C * cos_sig * (-1 + 2 * cos_2sm**2)))
        done = abs(lam_new - lam) < 1e-14
        lam = lam_new
        if done:
            break
    u2 = cos2_al * (a * a - b * b) / (b * b)
    A = 1 + u2 / 16384 * (4096 + u2 * (-768 + u2 * (320 - 175 * u2)))
    B = u2 / 1024 * (256 + u2 * (-128 + u2 * (74 - 47 * u2)))
    dsig = B * sin_sig * (cos_2sm + B / 4 * (cos_sig * (-1 + 2 * cos_2sm**2)
                                             - B / 6 * cos_2sm * (-3 + 4 * sin_sig**2) * (-3 + 4 * cos_2sm**2)))
    return b * A * (sig - dsig)


def distance_from_latlon(latitude, longitude):
    """Cumulative along-track distance per ping in nautical miles (commongrid/utils.py:208-231):
    distance from each ping to the NEXT one (shift(-1)), rows with a NaN position on either side
    dropped, cumulative sum, forward then backward fill."""
    df = pd.DataFrame({"latitude": np.asarray(latitude, float), "longitude": np.asarray(longitude, float)})
    df["latitude_prev"] = df["latitude"].shift(-1)
    df["longitude_prev"] = df["longitude"].shift(-1)
    nonan = df.dropna().copy()
    if len(nonan) == 0:
        raise ValueError("All lat/lon entries are NaN!")
    nonan["dist"] = [geodesic_m(r.latitude, r.longitude, r.latitude_prev, r.longitude_prev) / 1852.0
                     for r in nonan.itertuples()]
    df = df.join(nonan["dist"], how="left")
    df["dist"] = df["dist"].cumsum()
    df["dist"] = df["dist"].ffill().bfill()
    return df["dist"].values


def compute_raw_NASC(Sv, depth, dist, ping_time, r_edges, d_edges, skipna=True, closed="left"):
    """commongrid/utils.py:97-205 on (C,P,S) arrays -> (NASC (C, nd, nr), mean ping_time per distance bin)."""
    C, P, S = Sv.shape
    nd, nr = len(d_edges) - 1, len(r_edges) - 1
    with np.errstate(invalid="ignore", over="ignore"):
        sv = 10 ** (Sv / 10)
    idist = bin_index(dist, d_edges, closed)  # (P,)
    ir = bin_index(depth, r_edges, closed)  # (C,P,S)
    sv_mean = np.full((C, nd, nr), np.nan)
    h_num = np.zeros((C, nd, nr))
    with np.errstate(invalid="ignore"):
        dd = np.diff(depth, axis=2)  # label="lower": belongs to the upper (shallower) sample
    for c in range(C):
        flat = idist[:, None] * nr + ir[c]
        ok = (idist[:, None] >= 0) & (ir[c] >= 0)
        v = sv[c]
        use = ok & ~np.isnan(v) if skipna else ok
        ssum = np.bincount(flat[use], weights=v[use], minlength=nd * nr)
        n = np.bincount(flat[use], minlength=nd * nr)
        with np.errstate(invalid="ignore", divide="ignore"):
            sv_mean[c] = np.where(n > 0, ssum / np.where(n > 0, n, 1), np.nan).reshape(nd, nr)
        okh = ok[:, :-1] & ~np.isnan(dd[c])  # nansum
        h_num[c] = np.bincount(flat[:, :-1][okh], weights=dd[c][okh], minlength=nd * nr).reshape(nd, nr)
    h_denom = np.bincount(idist[idist >= 0], minlength=nd).astype(float)  # pings per distance bin
    with np.errstate(invalid="ignore", divide="ignore"):
        h_mean = h_num / h_denom[None, :, None]
        nasc = sv_mean * h_mean * 4 * np.pi * 1852**2
    t = np.asarray(ping_time).astype("datetime64[ns]").astype(np.int64).astype(np.float64)
    with warnings.catch_warnings(), np.errstate(invalid="ignore", divide="ignore"):
        warnings.simplefilter("ignore", RuntimeWarning)
        okt = idist >= 0
        t_mean = np.bincount(idist[okt], weights=t[okt], minlength=nd) / h_denom
    return nasc, t_mean


def compute_NASC(Sv, depth, latitude, longitude, ping_time, range_bin=10.0, dist_bin=0.5, skipna=True,
                 closed="left"):
    """commongrid/api.py:330-416 -> dict(NASC, distance, depth, ping_time, latitude, longitude)."""
    dist = distance_from_latlon(latitude, longitude)
    r_edges = np.arange(0, np.nanmax(depth) + range_bin, range_bin)
    d_edges = np.arange(0, np.nanmax(dist) + dist_bin, dist_bin)
    nasc, t_mean = compute_raw_NASC(Sv, depth, dist, ping_time, r_edges, d_edges, skipna, closed)
    idist = bin_index(dist, d_edges, closed)
    pos = {}
    for name, v in (("latitude", latitude), ("longitude", longitude)):  # utils.py:453-501 nanmean per bin
        v = np.asarray(v, float)
        ok = (idist >= 0) & ~np.isnan(v)
        s = np.bincount(idist[ok], weights=v[ok], minlength=len(d_edges) - 1)
        n = np.bincount(idist[ok], minlength=len(d_edges) - 1)
        with np.errstate(invalid="ignore", divide="ignore"):
            pos[name] = np.where(n > 0, s / n, np.nan)
    return dict(NASC=nasc, distance=d_edges[:-1], depth=r_edges[:-1], ping_time=t_mean, distance_nmi=dist, **pos)
