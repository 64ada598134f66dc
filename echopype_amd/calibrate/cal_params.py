"""Calibration parameters (host side, O(C*P)).

Mirrors /root/reference/echopype/calibrate/cal_params.py: CAL_PARAMS / EK80_DEFAULT_PARAMS
(:6-49), get_vend_cal_params_power (:261-324, pulse-length table lookup), get_cal_params_AZFP
(:327-362) and get_cal_params_EK (:365-522; user -> file -> defaults, BB interpolation over
``cal_frequency`` at the centre frequency).
"""
import numpy as np

from ..xr_lite import DataArray, DeviceArray, host_readable

CAL_PARAMS = {
    "EK60": ("sa_correction", "gain_correction", "equivalent_beam_angle", "angle_offset_alongship",
             "angle_offset_athwartship", "angle_sensitivity_alongship", "angle_sensitivity_athwartship",
             "beamwidth_alongship", "beamwidth_athwartship"),
    "EK80": ("sa_correction", "gain_correction", "equivalent_beam_angle", "angle_offset_alongship",
             "angle_offset_athwartship", "angle_sensitivity_alongship", "angle_sensitivity_athwartship",
             "beamwidth_alongship", "beamwidth_athwartship", "impedance_transducer",
             "impedance_transceiver", "receiver_sampling_frequency"),
    "AZFP": ("EL", "DS", "TVR", "VTX0", "equivalent_beam_angle", "Sv_offset"),
}

EK80_DEFAULT_PARAMS = {
    "impedance_transducer": 75,
    "impedance_transceiver": 1000,
    "receiver_sampling_frequency": {"default": 1500000, "GPT": 500000, "SBT": 50000, "WBAT": 1500000,
                                    "WBT TUBE": 1500000, "WBT MINI": 1500000, "WBT": 1500000,
                                    "WBT HP": 187500, "WBT LF": 93750},
}

_BEAM_NAME = {
    "angle_offset_alongship": "angle_offset_alongship",
    "angle_offset_athwartship": "angle_offset_athwartship",
    "angle_sensitivity_alongship": "angle_sensitivity_alongship",
    "angle_sensitivity_athwartship": "angle_sensitivity_athwartship",
    "beamwidth_alongship": "beamwidth_twoway_alongship",
    "beamwidth_athwartship": "beamwidth_twoway_athwartship",
    "equivalent_beam_angle": "equivalent_beam_angle",
}


class PulseTableParam(DataArray):
    """gain_correction / sa_correction as the reference returns them -- a (channel, ping_time) array looked up per ping
    in the Vendor_specific pulse-length table (cal_params.py:261-324) -- but LAZY: the calibrator hands the (C, K)
    table itself to the coefficient kernel (epa_power_coef_ek, EPA_PM_PULSE_TABLE), so the (C, P) array only comes
    into being when somebody reads it (``.values``, or the output dataset: one small kernel when the per-ping
    parameters live in HBM, the run-length NumPy evaluation otherwise)."""

    def __init__(self, tau_da, pl, tab, name):
        self._tau, self.pulse_length, self.table = tau_da, pl, tab
        self._data = None
        shape = tau_da.shape if tau_da.dims[0] == "channel" else tau_da.shape[::-1]
        self._shape = tuple(shape)
        self.dims = ("channel", "ping_time")
        from collections import OrderedDict

        self.coords = OrderedDict()
        self.attrs = {}
        self.name = name

    @property
    def materialized(self):
        return self._data is not None

    @property
    def on_device(self):
        """The device the per-ping pulse lengths live on as a (channel, ping_time) array, else None."""
        tau = self._tau.data
        if isinstance(tau, DeviceArray) and self._tau.dims[0] == "channel":
            return tau.tensor.device
        return None

    @property
    def shape(self):
        return self._shape

    @property
    def ndim(self):
        return 2

    @property
    def dtype(self):
        return np.dtype(np.float64)

    @property
    def data(self):
        if self._data is None:
            tau = self._tau.data
            if isinstance(tau, DeviceArray) and self._tau.dims[0] == "channel":
                import torch

                from .. import ops

                t = tau.tensor if tau.tensor.dtype == torch.float64 else tau.tensor.double()
                dev = lambda a: ops.to_device(np.ascontiguousarray(a, dtype=np.float64), device=t.device)  # noqa: E731
                self._data = DeviceArray(ops.pulse_table_lookup(t.contiguous(), dev(self.pulse_length), dev(self.table)))
            else:
                self._data = _lookup_host(np.asarray(self._tau.values, dtype=np.float64) if self._tau.dims[0] == "channel"
                                          else np.asarray(self._tau.values, dtype=np.float64).T, self.pulse_length, self.table)
        return self._data

    @data.setter
    def data(self, v):
        self._data = v

    @property
    def values(self):
        """Host values.  Per-ping pulse lengths that live in HBM WITH a host mirror (EchoData.to_device) are looked up
        on the host: a host-side consumer (the BB gain interpolation, get_cal_params_EK) then costs neither a kernel nor
        a wait for the GPU."""
        if self._data is None:
            tau = self._tau.data
            if isinstance(tau, DeviceArray) and host_readable(tau):
                if getattr(self, "_host_values", None) is None:
                    t = np.asarray(tau, dtype=np.float64)
                    self._host_values = _lookup_host(t if self._tau.dims[0] == "channel" else t.T, self.pulse_length,
                                                     self.table)
                return self._host_values
        return np.asarray(self.data)


def _lookup_host(tau, pl, tab):
    # The pulse length is piecewise constant along ping_time (usually constant): the (C, P, K) distance table of the
    # reference is evaluated only where some channel's tau changes, then repeated over the run -- same values,
    # O(C P) instead of O(C P K) temporaries (0.1 s per call at 4 x 500 000 pings otherwise).
    P = tau.shape[1]
    if P > 1:
        a, b = tau[:, 1:], tau[:, :-1]
        change = np.any((a != b) & ~(np.isnan(a) & np.isnan(b)), axis=0)
        starts = np.concatenate([[0], np.flatnonzero(change) + 1])
    else:
        starts = np.zeros(1, dtype=np.int64)
    tau_u = tau[:, starts]
    isnull = np.isnan(tau_u)
    diff = np.abs(tau_u[:, :, None] - pl[:, None, :])
    diff = np.where(np.isnan(diff), np.inf, diff)
    idx = np.argmin(diff, axis=2)
    out_u = np.where(isnull, np.nan, np.take_along_axis(tab, idx, axis=1))
    return np.repeat(out_u, np.diff(np.concatenate([starts, [P]])), axis=1) if starts.size > 1 else \
        np.repeat(out_u, P, axis=1)


def get_vend_cal_params_power(beam, vend, param):
    """argmin_k |transmit_duration_nominal - pulse_length[c,k]| lookup -> (C, P) array (lazy, see PulseTableParam)."""
    if param not in ("sa_correction", "gain_correction"):
        raise ValueError(f"Unknown parameter {param}")
    if param not in vend:
        raise ValueError(f"{param} does not exist in the Vendor_specific group!")
    pl = np.asarray(vend["pulse_length"].values, dtype=np.float64)
    tab = np.asarray(vend[param].values, dtype=np.float64)
    bch, vch = list(map(str, beam["channel"].values)), list(map(str, vend["channel"].values))
    if bch != vch:  # channel order differs between Vendor_specific and the beam group (:302-305)
        order = [vch.index(c) for c in bch]
        pl, tab = pl[order], tab[order]
    return PulseTableParam(beam["transmit_duration_nominal"], pl, tab, param)


def sanitize_user_cal_dict(sonar_type, user_dict, channel):
    """Allowed keys per sonar; scalars / per-channel lists / DataArrays (cal_params.py:85-162)."""
    if sonar_type not in CAL_PARAMS:
        raise ValueError(f"'sonar_type' has to be one of: {', '.join(CAL_PARAMS)}")
    channel = list(np.asarray(getattr(channel, "values", channel)))
    out = dict.fromkeys(CAL_PARAMS[sonar_type])
    for name, val in (user_dict or {}).items():
        if name not in out:
            continue
        if isinstance(val, DataArray):
            if "channel" not in val.coords and "cal_channel_id" not in val.coords:
                raise ValueError(f"{name} has to have either 'channel' or 'cal_channel_id' as a coordinate")
            out[name] = val
        elif isinstance(val, (int, float)):
            out[name] = DataArray(np.full(len(channel), float(val)), ("channel",), {"channel": np.asarray(channel)})
        elif isinstance(val, list):
            if len(val) != len(channel):
                raise ValueError("The lengths of param value and channel do not match!")
            out[name] = DataArray(np.asarray(val, dtype=np.float64), ("channel",), {"channel": np.asarray(channel)})
        else:
            raise ValueError(f"{name} has to be a scalar, list, or an xr.DataArray")
    return out


def _interp_freq(da_param, freq_center, alternative, channels, BB_factor=1.0):
    """Per channel: interpolate a (cal_channel_id, cal_frequency) table at the centre frequency, or
    fall back to ``alternative`` * BB_factor (cal_params.py:165-258)."""
    fc = np.asarray(freq_center, dtype=np.float64)  # (C,) or (C,P)
    alt = np.asarray(getattr(alternative, "values", alternative), dtype=np.float64)
    bbf = np.asarray(getattr(BB_factor, "values", BB_factor), dtype=np.float64)
    out = np.empty(fc.shape, dtype=np.float64)
    have = {}
    if da_param is not None and "cal_channel_id" in da_param.coords:
        ids = list(map(str, da_param.coords["cal_channel_id"]))
        have = {cid: i for i, cid in enumerate(ids)}
    for i, ch in enumerate(map(str, channels)):
        if ch in have:
            tab = np.asarray(da_param.values, dtype=np.float64)[have[ch]]
            freqs = np.asarray(da_param.coords["cal_frequency"], dtype=np.float64)
            if freqs.ndim == 2:
                freqs = freqs[have[ch]]
            ok = ~np.isnan(tab)
            out[i] = np.interp(fc[i], freqs[ok], tab[ok], left=np.nan, right=np.nan)
        else:
            a = alt if alt.ndim == 0 else alt[i]
            b = bbf if bbf.ndim == 0 else bbf[i]
            out[i] = a * b
    return DataArray(out, ("channel",) if out.ndim == 1 else ("channel", "ping_time"))


def get_cal_params_EK(waveform_mode, freq_center, beam, vend, user_dict, default_params=EK80_DEFAULT_PARAMS,
                      sonar_type="EK80"):
    if not isinstance(waveform_mode, str):
        raise TypeError("waveform_mode is not type string")
    if waveform_mode not in ("CW", "BB"):
        raise ValueError("waveform_mode must be 'CW' or 'BB'")
    channels = list(beam["channel"].values)
    out = sanitize_user_cal_dict(sonar_type, user_dict, beam["channel"])
    fc = np.asarray(getattr(freq_center, "values", freq_center), dtype=np.float64)
    fnom = np.asarray(beam["frequency_nominal"].values, dtype=np.float64)
    fn = fnom if fc.ndim == 1 else fnom[:, None]
    for p, v in out.items():
        if v is not None and "cal_channel_id" in v.coords:
            out[p] = _interp_freq(v, fc, np.nan, channels)

    def _fs():
        if "receiver_sampling_frequency" in vend and not np.isclose(vend["receiver_sampling_frequency"].values, 0).all():
            return vend["receiver_sampling_frequency"]
        tt = [str(t).upper() for t in vend["transceiver_type"].values]
        return DataArray(np.array([default_params["receiver_sampling_frequency"][t] for t in tt], float), ("channel",))

    for p, v in out.items():
        if v is not None:
            continue
        if p == "sa_correction":
            out[p] = get_vend_cal_params_power(beam, vend, p)
        elif p == "impedance_transceiver":
            out[p] = vend[p] if p in vend else default_params[p]
        elif p == "receiver_sampling_frequency":
            out[p] = _fs()
        elif waveform_mode == "CW":
            if p in _BEAM_NAME:
                out[p] = beam[_BEAM_NAME[p]] if _BEAM_NAME[p] in beam else None
            elif p == "gain_correction":
                out[p] = get_vend_cal_params_power(beam, vend, p)
            elif p == "impedance_transducer":
                out[p] = _interp_freq(vend[p] if p in vend else None, fc, default_params[p], channels)
            else:
                raise ValueError(f"{p} not in the defined set of calibration parameters.")
        else:  # BB
            if p in _BEAM_NAME and p != "equivalent_beam_angle":
                if p.startswith("angle_sensitivity"):
                    bbf = fc / fn
                elif p.startswith("beamwidth"):
                    bbf = fn / fc
                else:
                    bbf = 1.0
                alt = beam[_BEAM_NAME[p]].values if _BEAM_NAME[p] in beam else np.full(len(channels), np.nan)
                if np.ndim(alt) == 1 and fc.ndim == 2:
                    alt = np.asarray(alt)[:, None] * np.ones_like(fc)
                out[p] = _interp_freq(vend[p] if p in vend else None, fc, alt, channels, bbf)
            elif p == "equivalent_beam_angle":
                psi = np.asarray(beam[p].values, dtype=np.float64)
                psi = psi if fc.ndim == 1 else psi[:, None]
                v = psi + 20 * np.log10(fn / fc)
                out[p] = DataArray(v, ("channel",) if v.ndim == 1 else ("channel", "ping_time"))
            elif p == "gain_correction":
                alt = get_vend_cal_params_power(beam, vend, p).values
                if fc.ndim == 1:
                    alt = alt[:, 0]
                out[p] = _interp_freq(vend["gain"] if "gain" in vend else None, fc, alt, channels)
            elif p == "impedance_transducer":
                out[p] = _interp_freq(vend[p] if p in vend else None, fc, default_params[p], channels)
            else:
                raise ValueError(f"{p} not in the defined set of calibration parameters.")
    return out


def get_cal_params_AZFP(beam, vend, user_dict):
    out = sanitize_user_cal_dict("AZFP", user_dict, beam["channel"])
    for p, v in out.items():
        if v is None:
            out[p] = beam[p] if p == "equivalent_beam_angle" else vend[p]
    return out
