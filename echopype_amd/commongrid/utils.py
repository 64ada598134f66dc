"""Host helpers of compute_MVBS (mirrors /root/reference/echopype/commongrid/utils.py:283-450,
654-698 and the bin construction of commongrid/api.py:108-128).  O(P) work only."""
import re

import numpy as np

_UNIT_NS = {"ns": 1, "n": 1, "us": 10**3, "u": 10**3, "ms": 10**6, "l": 10**6, "s": 10**9, "sec": 10**9,
            "second": 10**9, "seconds": 10**9, "min": 60 * 10**9, "t": 60 * 10**9, "m": 60 * 10**9,
            "minute": 60 * 10**9, "minutes": 60 * 10**9, "h": 3600 * 10**9, "hr": 3600 * 10**9,
            "hour": 3600 * 10**9, "hours": 3600 * 10**9, "d": 86400 * 10**9, "day": 86400 * 10**9,
            "days": 86400 * 10**9}
_LABEL = {1: ("n", "nanosecond"), 10**3: ("us", "microsecond"), 10**6: ("ms", "millisecond"),
          10**9: ("s", "second"), 60 * 10**9: ("min", "minute"), 3600 * 10**9: ("h", "hour"),
          86400 * 10**9: ("d", "day")}


def _parse_x_bin(x_bin, x_label="range_bin"):
    """'10m' -> 10.0 (utils.py:305-377; same error types and messages)."""
    info = {"range_bin": ("Range bin", "m", "10m", "meters", r"([\d+]*[.,]{0,1}[\d+]*)(\s+)?(m)"),
            "dist_bin": ("Distance bin", "nmi", "0.5nmi", "nautical miles", r"([\d+]*[.,]{0,1}[\d+]*)(\s+)?(nmi)")}
    if x_label not in info:
        raise KeyError(f"x_label must be one of {list(info)}")
    name, _, ex, unit_label, pattern = info[x_label]
    if not isinstance(x_bin, str):
        raise TypeError("'x_bin' must be a string")
    m = re.match(pattern, x_bin.strip().lower())
    if m is None:
        raise ValueError(f"{name} must be in {unit_label} (e.g., '{ex}').")
    return float(m.group(1))


def timedelta_ns(ping_time_bin):
    """'20s' / '1min' / '0.5h' -> integer nanoseconds (what pd.Timedelta(str).value returns)."""
    m = re.fullmatch(r"\s*([0-9]*\.?[0-9]+)\s*([A-Za-z]+)\s*", ping_time_bin)
    if m is None or m.group(2).lower() not in _UNIT_NS and m.group(2) not in ("T", "L", "U", "N", "D", "H", "S"):
        raise ValueError(f"invalid ping_time_bin {ping_time_bin!r}")
    unit = m.group(2) if m.group(2).lower() in _UNIT_NS else m.group(2).lower()
    ns = float(m.group(1)) * _UNIT_NS[unit.lower()]
    if ns <= 0 or ns != int(ns):
        raise ValueError(f"invalid ping_time_bin {ping_time_bin!r}")
    return int(ns)


def ping_time_bin_parsing_and_conversion(ping_time_bin):
    """(value, unit label) of the most granular unit, e.g. '20s' -> (20, 'second') (utils.py:654-698)."""
    ns = timedelta_ns(ping_time_bin)
    for unit_ns in sorted(_LABEL, reverse=True):
        if ns % unit_ns == 0:
            return ns // unit_ns, _LABEL[unit_ns][1]
    return ns, "nanosecond"


def resample_edges(ping_time, ping_time_bin, sorted_valid=False):
    """Bin edges of ``ping_time.resample(ping_time=bin)`` plus one trailing edge (api.py:118-124).

    pandas anchors fixed-frequency bins at midnight of the first timestamp's day
    (origin='start_day'), left-closed and left-labelled; bins run contiguously to the one holding
    the last timestamp.  Returned as int64 ns: (first_edge, step, n_bins).
    """
    t = np.asarray(ping_time).astype("datetime64[ns]", copy=False).view(np.int64)
    dt = timedelta_ns(ping_time_bin)
    if sorted_valid and t.size:  # the caller has checked: non-decreasing, no NaT -- the ends are the extremes
        first, last = int(t[0]), int(t[-1])
        day = 86400 * 10**9
        origin = (first // day) * day
        e0 = origin + ((first - origin) // dt) * dt
        return e0, dt, int((last - e0) // dt + 1)
    first = int(t.min()) if t.size else 0
    if t.size == 0 or first == np.iinfo(np.int64).min:  # NaT present (INT64_MIN): drop them
        t = t[t != np.iinfo(np.int64).min]
        if t.size == 0:
            raise ValueError("ping_time holds no valid timestamps")
        first = int(t.min())
    last = int(t.max())
    day = 86400 * 10**9
    origin = (first // day) * day
    e0 = origin + ((first - origin) // dt) * dt
    n = (last - e0) // dt + 1
    return e0, dt, int(n)


def coarsen_time_mean(ping_time, n):
    """``ping_time`` labels of ``da.coarsen(ping_time=n, boundary="pad")``: xarray reduces the coordinates of a
    coarsened dimension with its default ``coord_func="mean"`` -- per window the NaT-skipping float mean of the
    ns offsets from the earliest timestamp, truncated to whole ns (api.py:217-221 keeps these labels).
    Host O(P)."""
    t = np.asarray(ping_time).astype("datetime64[ns]")
    if t.size == 0:
        return t
    valid = ~np.isnat(t)
    if not valid.any():
        return np.full(-(-t.size // n), np.datetime64("NaT", "ns"))
    off = t[valid].min()
    rel = np.where(valid, (t - off).astype(np.int64).astype(np.float64), np.nan)
    rel = np.pad(rel, (0, (-t.size) % n), constant_values=np.nan).reshape(-1, n)
    cnt = np.sum(~np.isnan(rel), axis=1)
    mean = np.divide(np.nansum(rel, axis=1), cnt, out=np.zeros(len(cnt)), where=cnt > 0)
    out = off + mean.astype(np.int64).astype("timedelta64[ns]")
    out[cnt == 0] = np.datetime64("NaT", "ns")
    return out


def _setup_and_validate(ds_Sv, range_var="echo_range", range_bin=None, closed="left", required_data_vars=None):
    """Argument checks of compute_MVBS (utils.py:380-450)."""
    if range_var not in ["echo_range", "depth"]:
        raise ValueError("range_var must be one of 'echo_range' or 'depth'.")
    required = set((required_data_vars or []) + [range_var])
    if not all(v in ds_Sv.variables for v in required):
        raise ValueError(f"Input Sv dataset must contain all of the following variables: {required}")
    if not isinstance(range_bin, str):
        raise TypeError("range_bin must be a string")
    range_bin = _parse_x_bin(range_bin, "range_bin")
    if closed not in ["right", "left"]:
        raise ValueError(f"{closed} is not a valid option. Options are 'left' or 'right'.")
    return ds_Sv, range_bin


def _set_MVBS_attrs(ds):
    """utils.py:234-262."""
    ds.coords["ping_time"].attrs = {"long_name": "Ping time", "standard_name": "time", "axis": "T"}
    ds.data_vars["Sv"].attrs = {"long_name": "Mean volume backscattering strength (MVBS, mean Sv re 1 m-1)",
                                "units": "dB", "actual_range": None}
    ds.data_vars["Sv"].attrs.pop("actual_range")


# ---- compute_NASC host helpers (utils.py:208-231) ------------------------------------
_WGS84_A = 6378137.0
_WGS84_F = 1 / 298.257223563


def assign_actual_range(ds_MVBS):
    """Post-computation ``actual_range`` attribute of an MVBS dataset: [min, max] of ``Sv`` rounded to 2 digits
    (commongrid/utils.py:631-651; compute_MVBS itself does not set it).  One reduction on the device."""
    from .. import ops
    from ..xr_lite import DeviceArray

    d = ds_MVBS["Sv"].data
    if isinstance(d, DeviceArray):
        lo, hi = ops.nanminmax(d.tensor)
    else:
        a = np.asarray(d, dtype=np.float64)
        lo, hi = np.nanmin(a), np.nanmax(a)
    return ds_MVBS.assign_attrs({"actual_range": [round(float(lo), 2), round(float(hi), 2)]})


_GEODESIC_ON_DEVICE = 8192  # pings from which get_distance_from_latlon takes the steps from the device kernel


def geodesic_distance_m(lat1, lon1, lat2, lon2):
    """Geodesic length in metres on WGS-84 between arrays of points (degrees), the quantity the
    reference takes from ``geopy.distance.distance`` (utils.py:219-225).  Vincenty's inverse
    iteration, vectorised; coincident points give 0."""
    a, f = _WGS84_A, _WGS84_F
    b = (1 - f) * a
    lat1, lon1, lat2, lon2 = (np.asarray(v, dtype=np.float64) for v in (lat1, lon1, lat2, lon2))
    U1 = np.arctan((1 - f) * np.tan(np.radians(lat1)))
    U2 = np.arctan((1 - f) * np.tan(np.radians(lat2)))
    L = np.radians(lon2 - lon1)
    sU1, cU1, sU2, cU2 = np.sin(U1), np.cos(U1), np.sin(U2), np.cos(U2)
    lam = L.copy()
    with np.errstate(invalid="ignore", divide="ignore"):
        for _ in range(200):
            sl, cl = np.sin(lam), np.cos(lam)
            sin_sig = np.hypot(cU2 * sl, cU1 * sU2 - sU1 * cU2 * cl)
            cos_sig = sU1 * sU2 + cU1 * cU2 * cl
            sig = np.arctan2(sin_sig, cos_sig)
            sin_al = np.where(sin_sig == 0, 0.0, cU1 * cU2 * sl / sin_sig)
            cos2_al = 1 - sin_al**2
            cos_2sm = np.where(cos2_al == 0, 0.0, cos_sig - 2 * sU1 * sU2 / cos2_al)
            C = f / 16 * cos2_al * (4 + f * (4 - 3 * cos2_al))
            lam_new = L + (1 - C) * f * sin_al * (sig + C * sin_sig * (cos_2sm + C * cos_sig * (-1 + 2 * cos_2sm**2)))
            done = np.all(np.abs(lam_new - lam) < 1e-14)
            lam = lam_new
            if done:
                break
        u2 = cos2_al * (a * a - b * b) / (b * b)
        A = 1 + u2 / 16384 * (4096 + u2 * (-768 + u2 * (320 - 175 * u2)))
        B = u2 / 1024 * (256 + u2 * (-128 + u2 * (74 - 47 * u2)))
        dsig = B * sin_sig * (cos_2sm + B / 4 * (cos_sig * (-1 + 2 * cos_2sm**2)
                                                 - B / 6 * cos_2sm * (-3 + 4 * sin_sig**2) * (-3 + 4 * cos_2sm**2)))
        s = b * A * (sig - dsig)
    return np.where(sin_sig == 0, 0.0, s)


def get_distance_from_latlon(ds_Sv):
    """Cumulative along-track distance of every ping in nautical miles (utils.py:208-231): distance to
    the NEXT ping, pairs with a NaN position dropped, cumulative sum, forward then backward fill."""
    lat = np.asarray(ds_Sv["latitude"].values, dtype=np.float64)
    lon = np.asarray(ds_Sv["longitude"].values, dtype=np.float64)
    P = lat.shape[0]
    lat2, lon2 = np.r_[lat[1:], np.nan], np.r_[lon[1:], np.nan]
    ok = ~(np.isnan(lat) | np.isnan(lon) | np.isnan(lat2) | np.isnan(lon2))
    if not ok.any():
        raise ValueError("All lat/lon entries are NaN!")
    step = np.zeros(P)
    if P >= _GEODESIC_ON_DEVICE:  # a long track: one launch (30 ms of NumPy per 100 000 pings otherwise)
        from .. import ops

        dev = ops.geodesic_steps(ops.to_device(lat), ops.to_device(lon)).cpu().numpy()
        step[ok] = dev[ok] / 1852.0
    else:
        step[ok] = geodesic_distance_m(lat[ok], lon[ok], lat2[ok], lon2[ok]) / 1852.0
    dist = np.where(ok, np.cumsum(step), np.nan)  # pandas cumsum skips NaN rows but keeps them NaN
    idx = np.where(ok, np.arange(P), -1)
    last = np.maximum.accumulate(idx)  # forward fill
    dist = np.where(last >= 0, dist[np.maximum(last, 0)], np.nan)
    first = np.flatnonzero(ok)[0]
    dist[:first] = dist[first]  # backward fill of the leading gap
    return dist
