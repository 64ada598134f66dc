"""Environmental parameters for calibration (host side, O(C*P)).

Mirrors the behaviour of /root/reference/echopype/calibrate/env_params.py:
  harmonize_env_param_time :24-71   (time1 -> ping_time; one timestamp squeezes, else linear
                                     interpolation with extrapolation, utils/align.py:54-60)
  sanitize_user_env_dict   :74-157  (allowed keys, list -> per-channel, type errors)
  get_env_params_AZFP      :160-221
  get_env_params_EK        :224-353 (user -> data file -> formula precedence)
Parameters are plain numpy values of shape (), (C,) or (C, P).
"""
import numpy as np

from ..utils import uwa
from ..xr_lite import DataArray, DeviceArray

ENV_PARAMS = ("sound_speed", "sound_absorption", "temperature", "salinity", "pressure", "pH",
              "formula_sound_speed", "formula_absorption")


def _interp_time(values, t_src, t_dst):
    """Linear interpolation along the last axis with linear extrapolation."""
    x = t_src.astype("datetime64[ns]").astype(np.int64).astype(np.float64)
    xq = t_dst.astype("datetime64[ns]").astype(np.int64).astype(np.float64)
    hi = np.clip(np.searchsorted(x, xq, side="left"), 1, x.size - 1)
    lo = hi - 1
    slope = (values[..., hi] - values[..., lo]) / (x[hi] - x[lo])
    return slope * (xq - x[lo]) + values[..., lo]


def _same_buffer(a, b):
    """The two time axes are one and the same array (the usual EK60 file: Environment.time1 is ping_time)."""
    return a is b or (a.shape == b.shape and a.dtype == b.dtype and a.__array_interface__["data"][0] ==
                      b.__array_interface__["data"][0] and a.strides == b.strides)


def harmonize_env_param_time(p, ping_time=None):
    """Bring a parameter with a ``time1`` dimension onto ``ping_time``; anything else passes through."""
    if not isinstance(p, DataArray) or "time1" not in p.dims:
        return p
    if isinstance(p.data, DeviceArray):  # per-ping parameters resident in HBM (EchoData.to_device)
        t1 = np.asarray(p.coords["time1"]) if "time1" in p.coords else None
        pt = None if ping_time is None else np.asarray(getattr(ping_time, "values", ping_time))
        dims = tuple("ping_time" if d == "time1" else d for d in p.dims)
        if p.sizes["time1"] == 1:
            ax1 = p.dims.index("time1")
            return DataArray(DeviceArray(p.data.tensor.select(ax1, 0).contiguous()), tuple(d for d in p.dims if d != "time1"),
                             {d: p.coords[d] for d in p.dims if d != "time1" and d in p.coords})
        if t1 is not None and pt is not None and t1.shape == pt.shape and (_same_buffer(t1, pt) or np.array_equal(t1, pt)):
            coords = {d: p.coords[d] for d in p.dims if d != "time1" and d in p.coords}
            coords["ping_time"] = pt
            return DataArray(p.data, dims, coords)
        p = DataArray(np.asarray(p.values), p.dims, p.coords, p.attrs, p.name)  # anything else: the host path
    ax = p.dims.index("time1")
    vals = np.moveaxis(np.asarray(p.values, dtype=np.float64), ax, -1)
    dims = tuple(d for d in p.dims if d != "time1")
    t1 = np.asarray(p.coords["time1"]) if "time1" in p.coords else None
    if vals.shape[-1] == 1:
        return DataArray(vals[..., 0], dims, {d: p.coords[d] for d in dims if d in p.coords})
    finite = ~np.isnan(vals).reshape(-1, vals.shape[-1]).all(axis=0)
    if finite.sum() == 1 and vals.ndim == 1:  # only one non-NaN along time1 (:57-58)
        return DataArray(vals[finite][0], ())
    if ping_time is None:
        raise ValueError(f"ping_time needs to be provided for comparison or interpolating {p.name}")
    pt = np.asarray(getattr(ping_time, "values", ping_time))
    if not finite.all():
        vals, t1 = vals[..., finite], t1[finite]
    if t1.shape == pt.shape and (_same_buffer(t1, pt) or np.array_equal(t1, pt)):
        out = vals
    else:
        out = _interp_time(vals, t1, pt)
    coords = {d: p.coords[d] for d in dims if d in p.coords}
    coords["ping_time"] = pt
    return DataArray(out, dims + ("ping_time",), coords)


def param2array(p_val, channel):
    """list -> per-channel array with the reference's checks (cal_params.py:52-82)."""
    if isinstance(p_val, (int, float)):
        return np.float64(p_val)
    if isinstance(p_val, list):
        if len(p_val) != len(channel):
            raise ValueError("The lengths of param value and channel do not match!")
        return np.asarray(p_val, dtype=np.float64)
    raise ValueError("p_val has to be one of int, float, or list")


def sanitize_user_env_dict(user_dict, channel):
    channel = list(np.asarray(getattr(channel, "values", channel)))
    out = dict.fromkeys(ENV_PARAMS)
    for name, val in (user_dict or {}).items():
        if name not in out:
            continue
        if name == "sound_absorption" and not isinstance(val, (DataArray, list)):
            raise ValueError("The 'sound_absorption' parameter has to be a list or an xr.DataArray, "
                             "with 'channel' as an coordinate.")
        if isinstance(val, DataArray):
            if "channel" not in val.coords:
                raise ValueError(f"{name} has to have 'channel' as a coordinate")
            if sorted(map(str, val.coords["channel"])) != sorted(map(str, channel)):
                raise ValueError(f"The 'channel' coordinate of {name} has to match that of the data to be calibrated")
            out[name] = val
        elif isinstance(val, (int, float, str)):
            out[name] = val
        elif isinstance(val, list):
            out[name] = DataArray(param2array(val, channel), ("channel",), {"channel": np.asarray(channel)})
        else:
            raise ValueError(f"{name} has to be a scalar, list, or an xr.DataArray")
    return out


def _val(p):
    return p.values if isinstance(p, DataArray) else p


def get_env_params_EK(sonar_type, beam, env, user_dict=None, freq=None):
    """EK60/EK80 env params: user -> file -> formula (env_params.py:224-353)."""
    if sonar_type not in ("EK60", "EK80"):
        raise ValueError("'sonar_type' has to be 'EK60' or 'EK80'")
    if sonar_type == "EK80":
        if freq is None:
            raise ValueError("'freq' is required for calibrating EK80-style data.")
    else:
        freq = beam["frequency_nominal"]
    user_dict = user_dict or {}
    out = sanitize_user_env_dict(user_dict, beam["channel"])
    if out["formula_absorption"] not in (None, "AM", "FG"):
        raise ValueError("'formula_absorption' has to be None, 'FG' or 'AM' for EK echosounders.")
    if out["formula_sound_speed"] not in (None, "Mackenzie"):
        raise ValueError("'formula_absorption' has to be None or 'Mackenzie' for EK echosounders.")
    tspa = all(out[p] is not None for p in ("temperature", "salinity", "pressure", "pH"))
    if not tspa and sonar_type == "EK80":
        for pu, pd in zip(("temperature", "salinity", "pressure", "pH"),
                          ("temperature", "salinity", "depth", "acidity")):
            out[pu] = user_dict.get(pu, env[pd])

    # The reference evaluates the sound-speed / absorption formulas on the parameters' NATIVE time axis (``time1`` of the
    # Environment group) and brings the RESULTS onto ping_time afterwards (env_params.py:300-351): with several
    # Environment timestamps interp(f(x)) != f(interp(x)).  ``native`` keeps the un-harmonised values for that.
    native = dict(out)
    if out["sound_speed"] is None and not tspa:
        native["sound_speed"] = env["sound_speed_indicative"]
    for p in ("temperature", "salinity", "pressure", "pH", "sound_speed", "sound_absorption"):
        out[p] = harmonize_env_param_time(out[p], beam["ping_time"]) if isinstance(out[p], DataArray) else out[p]

    def bc(p):  # broadcast helper: (C,) against (C,P)/(P,)
        v = _val(p)
        if isinstance(p, DataArray) and p.dims == ("channel",):
            return np.asarray(v)[:, None]
        return v

    def on_time1(names):
        """The shared multi-valued, NaN-free ``time1`` axis of the named native parameters, or None (then the
        harmonised values are used: one timestamp, no time axis at all, or axes that differ)."""
        t1 = None
        for n in names:
            v = native.get(n)
            if isinstance(v, DataArray) and "channel" in v.dims:
                return None  # a per-channel parameter: the harmonised branch broadcasts it (bc) against (C, P)
            if isinstance(v, DataArray) and "time1" in v.dims and v.sizes["time1"] > 1:
                if v.dims != ("time1",) or np.isnan(np.asarray(v.values, dtype=np.float64)).any():
                    return None
                c = np.asarray(v.coords["time1"])
                if t1 is not None and not np.array_equal(t1, c):
                    return None
                t1 = c
        return t1

    def brackets(t1):
        """Per ping: the two time1 samples linear interpolation / extrapolation uses, and the weight of the upper."""
        x = t1.astype("datetime64[ns]").astype(np.int64).astype(np.float64)
        xq = np.asarray(beam["ping_time"].values).astype("datetime64[ns]").astype(np.int64).astype(np.float64)
        hi = np.clip(np.searchsorted(x, xq, side="left"), 1, x.size - 1)
        lo = hi - 1
        return lo, hi, (xq - x[lo]) / (x[hi] - x[lo])

    def at(name, idx):
        v = native[name]
        if isinstance(v, DataArray) and v.dims == ("time1",) and v.sizes["time1"] > 1:
            return np.asarray(v.values, dtype=np.float64)[idx]
        h = out[name] if name in out else harmonize_env_param_time(v, beam["ping_time"])
        return bc(h)

    if out["sound_speed"] is None:
        if not tspa:
            out["sound_speed"] = harmonize_env_param_time(env["sound_speed_indicative"], beam["ping_time"])
            out.pop("formula_sound_speed")
        else:
            out["formula_sound_speed"] = out["formula_sound_speed"] or "Mackenzie"
            t1 = on_time1(("temperature", "salinity", "pressure"))
            if t1 is None:
                out["sound_speed"] = uwa.calc_sound_speed(
                    temperature=_val(out["temperature"]), salinity=_val(out["salinity"]),
                    pressure=_val(out["pressure"]), formula_source=out["formula_sound_speed"])
            else:  # formula on time1, then onto ping_time
                ss1 = uwa.calc_sound_speed(temperature=at("temperature", slice(None)), salinity=at("salinity", slice(None)),
                                           pressure=at("pressure", slice(None)), formula_source=out["formula_sound_speed"])
                native["sound_speed"] = DataArray(np.asarray(ss1, dtype=np.float64), ("time1",), {"time1": t1})
                out["sound_speed"] = harmonize_env_param_time(native["sound_speed"], beam["ping_time"])
    else:
        out.pop("formula_sound_speed")
    if out["sound_absorption"] is None:
        if not tspa and sonar_type != "EK80":
            out["sound_absorption"] = harmonize_env_param_time(env["absorption_indicative"], beam["ping_time"])
            out.pop("formula_absorption")
        else:
            out["formula_absorption"] = out["formula_absorption"] or "FG"
            f = np.asarray(_val(freq), dtype=np.float64)
            names = ("temperature", "salinity", "pressure", "pH", "sound_speed")
            t1 = on_time1(names)
            if t1 is not None:  # formula at the two bracketing Environment timestamps of every ping, then interpolated
                lo, hi, w = brackets(t1)
                fq = f[:, None] if f.ndim == 1 else f
                ab_lo, ab_hi = (np.asarray(uwa.calc_absorption(
                    frequency=fq, temperature=at("temperature", i), salinity=at("salinity", i), pressure=at("pressure", i),
                    pH=at("pH", i), sound_speed=at("sound_speed", i), formula_source=out["formula_absorption"]),
                    dtype=np.float64) for i in (lo, hi))
                ab = ab_lo + (ab_hi - ab_lo) * w
                out["sound_absorption"] = DataArray(np.ascontiguousarray(ab), ("channel", "ping_time"))
            else:
                ss = _val(out["sound_speed"])
                if isinstance(out["sound_speed"], DataArray) and out["sound_speed"].dims == ("ping_time",) and f.ndim == 1:
                    f, ss = f[:, None], np.asarray(ss)[None, :]
                elif isinstance(out["sound_speed"], DataArray) and out["sound_speed"].dims == ("channel",) and f.ndim == 2:
                    ss = np.asarray(ss)[:, None]
                ab = uwa.calc_absorption(frequency=f, temperature=bc(out["temperature"]), salinity=bc(out["salinity"]),
                                         pressure=bc(out["pressure"]), pH=bc(out["pH"]), sound_speed=ss,
                                         formula_source=out["formula_absorption"])
                ab = np.asarray(ab, dtype=np.float64)
                dims = ("channel",) if ab.ndim == 1 else ("channel", "ping_time")
                out["sound_absorption"] = DataArray(ab, dims)
    else:
        out.pop("formula_absorption")
    if not ("formula_sound_speed" in out or "formula_absorption" in out):
        for p in ("temperature", "salinity", "pressure", "pH"):
            out.pop(p)
    return out


def get_env_params_AZFP(echodata, user_dict=None):
    """AZFP env params (env_params.py:160-221): salinity & pressure must come from the user."""
    beam = echodata["Sonar/Beam_group1"]
    user_dict = user_dict or {}
    out = sanitize_user_env_dict(user_dict, beam["channel"])
    out.pop("pH")
    if out.get("salinity") is None or out.get("pressure") is None:
        raise ReferenceError("Please supply both salinity and pressure in env_params.")
    if out["temperature"] is None:
        out["temperature"] = echodata["Environment"]["temperature"]
    out["formula_sound_speed"] = out["formula_sound_speed"] or "AZFP"
    out["formula_absorption"] = out["formula_absorption"] or "AZFP"
    for p in ("temperature", "sound_speed", "sound_absorption"):
        if isinstance(out[p], DataArray):
            out[p] = harmonize_env_param_time(out[p], beam["ping_time"])
    T = _val(out["temperature"])
    if out["sound_speed"] is None:
        out["sound_speed"] = uwa.calc_sound_speed(temperature=T, salinity=_val(out["salinity"]),
                                                  pressure=_val(out["pressure"]),
                                                  formula_source=out["formula_sound_speed"])
        if np.ndim(out["sound_speed"]) == 1:
            out["sound_speed"] = DataArray(out["sound_speed"], ("ping_time",))
    if out["sound_absorption"] is None:
        f = np.asarray(beam["frequency_nominal"].values, dtype=np.float64)
        Tb = np.asarray(T)[None, :] if np.ndim(T) == 1 else T
        ab = uwa.calc_absorption(frequency=f[:, None] if np.ndim(T) == 1 else f, temperature=Tb,
                                 salinity=_val(out["salinity"]), pressure=_val(out["pressure"]),
                                 formula_source=out["formula_absorption"])
        ab = np.asarray(ab)
        out["sound_absorption"] = DataArray(ab, ("channel",) if ab.ndim == 1 else ("channel", "ping_time"))
    return out
