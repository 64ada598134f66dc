"""Oracle: Sv / TS calibration chains for EK60, EK80 and AZFP (test infrastructure).

NumPy fp64 restatement of
  /root/reference/echopype/calibrate/range.py            (echo_range, TVG range shift)
  /root/reference/echopype/calibrate/calibrate_ek.py     (power & complex sample chains)
  /root/reference/echopype/calibrate/calibrate_azfp.py   (AZFP chain)
  /root/reference/echopype/calibrate/cal_params.py:261-324 (pulse-length table lookup)
  /root/reference/echopype/calibrate/env_params.py:24-71 + utils/align.py:54-60 (time harmonisation)

The functions keep the reference's pass structure: each arithmetic step is a
whole-array NumPy temporary in float64, evaluated in the reference's order.
Array convention: ``(channel, ping_time, range_sample[, beam])``; per-ping parameters
are (C, P); per-channel parameters are (C,).
"""
import numpy as np

from . import ek80

__all__ = [
    "vend_cal_params_power",
    "harmonize_time",
    "range_ek",
    "range_azfp",
    "tvg_range_ek",
    "cal_power_ek",
    "cal_complex_ek80",
    "cal_azfp",
    "b_theta_phi_m",
]


def _cp(a, C, P):
    """Broadcast scalar / (C,) / (C,P) parameter to (C, P, 1) float64."""
    a = np.asarray(a, dtype=np.float64)
    if a.ndim == 0:
        a = np.full((C, P), float(a))
    elif a.ndim == 1:
        a = np.broadcast_to(a[:, None], (C, P))
    return a[:, :, None]


def vend_cal_params_power(tau_nominal, pulse_length, table):
    """Pulse-length table lookup.  (cal_params.py:261-324)

    tau_nominal (C,P) may contain NaN; pulse_length, table (C,K).
    idx = argmin_k |tau - pulse_length[c,k]| (first minimum, :291-293); NaN tau -> NaN (:316).
    """
    tau_nominal = np.asarray(tau_nominal, dtype=np.float64)
    isnull = np.isnan(tau_nominal)
    diff = np.abs(tau_nominal[:, :, None] - np.asarray(pulse_length, dtype=np.float64)[:, None, :])
    diff = np.where(isnull[:, :, None], 0.0, diff)  # idxmin skips NaN; value is masked below
    idx = np.argmin(diff, axis=2)
    out = np.take_along_axis(np.asarray(table, dtype=np.float64), idx, axis=1)
    return np.where(isnull, np.nan, out)


def harmonize_time(values, time1, ping_time):
    """Bring an Environment-group series onto ping_time.  (env_params.py:24-71, align.py:33-61)

    values (..., T) along time1.  One timestamp -> squeezed; identical axes -> renamed;
    otherwise linear interpolation with linear extrapolation (scipy interp1d
    fill_value="extrapolate").
    """
    values = np.asarray(values, dtype=np.float64)
    time1 = np.asarray(time1)
    ping_time = np.asarray(ping_time)
    if time1.size == 1:
        return values[..., 0]
    if time1.shape == ping_time.shape and np.array_equal(time1, ping_time):
        return values
    x = time1.astype("datetime64[ns]").astype(np.int64).astype(np.float64)
    xq = ping_time.astype("datetime64[ns]").astype(np.int64).astype(np.float64)
    # xarray interp -> scipy interp1d(linear, extrapolate): slope of the bracketing
    # (or outermost) pair
    hi = np.clip(np.searchsorted(x, xq, side="left"), 1, x.size - 1)
    lo = hi - 1
    slope = (values[..., hi] - values[..., lo]) / (x[hi] - x[lo])
    return slope * (xq - x[lo]) + values[..., lo]


def range_ek(backscatter_r, sample_interval, sound_speed):
    """echo_range for EK60/EK80.  (range.py:138-148)

    R = range_sample * sample_interval * sound_speed / 2, NaN where backscatter_r
    (beam 0 for complex data) is NaN.
    """
    bs = backscatter_r[..., 0] if backscatter_r.ndim == 4 else backscatter_r
    C, P, S = bs.shape
    s = np.arange(S, dtype=np.int64)[None, None, :]
    R = s * _cp(sample_interval, C, P) * _cp(sound_speed, C, P) / 2
    return np.where(~np.isnan(bs), R, np.nan)


def tvg_range_ek(R, sonar, sample_interval, sound_speed, tau_nominal, gpt=None):
    """Range used for TVG.  (range.py:160-201)

    EK60/ES70: R - 2*si*c/2.  EK80/ES80/EA640: R - c*tau/4, and for GPT channels the
    Ex60 shift is subtracted *in addition* (:197-199).
    """
    C, P, _ = R.shape
    si, c = _cp(sample_interval, C, P), _cp(sound_speed, C, P)
    ex60 = 2 * si * c / 2
    if sonar in ("EK60", "ES70"):
        return R - ex60
    if sonar in ("EK80", "ES80", "EA640"):
        out = R - c * _cp(tau_nominal, C, P) / 4
        if gpt is not None and np.any(gpt):
            g = np.asarray(gpt, dtype=bool)
            out[g] = out[g] - ex60[g]
        return out
    raise ValueError("The specified sonar_model is not supported!")


def cal_power_ek(
    backscatter_r,
    *,
    sonar,
    cal_type,
    sample_interval,
    sound_speed,
    absorption,
    transmit_power,
    tau_nominal,
    gain,
    sa_correction,
    psi,
    f_nominal,
    tau_eff,
    gpt=None,
):
    """Power-sample Sv / TS for EK60 and EK80.  (calibrate_ek.py:79-206)

    tau_eff (C,): effective pulse length already resolved by the caller -- for EK60 the
    reference always ends at tau_nominal[c, ping 0] (:134,:140-151).
    Returns (out, echo_range) in float64.
    """
    C, P, S = backscatter_r.shape
    cw = _cp(sound_speed, C, P)
    R = range_ek(backscatter_r, sample_interval, sound_speed)
    wavelength = cw / _cp(f_nominal, C, P)  # :98
    Rt = tvg_range_ek(R, sonar, sample_interval, sound_speed, tau_nominal, gpt)
    with np.errstate(invalid="ignore", divide="ignore"):
        Rt = np.where(Rt > 0, Rt, np.nan)  # :107
        spreading = 20 * np.log10(Rt)  # :109
        absorb = 2 * _cp(absorption, C, P) * Rt  # :110
        if cal_type == "Sv":
            CSv = (
                10 * np.log10(_cp(transmit_power, C, P))
                + 2 * _cp(gain, C, P)
                + _cp(psi, C, P)
                + 10 * np.log10(wavelength**2 * _cp(tau_eff, C, P) * cw / (32 * np.pi**2))
            )  # :154-162
            out = backscatter_r + spreading + absorb - CSv - 2 * _cp(sa_correction, C, P)  # :165-171
        elif cal_type == "TS":
            CSp = (
                10 * np.log10(_cp(transmit_power, C, P))
                + 2 * _cp(gain, C, P)
                + 10 * np.log10(wavelength**2 / (16 * np.pi**2))
            )  # :176-181
            out = backscatter_r + spreading * 2 + absorb - CSp  # :184
        else:
            raise ValueError(cal_type)
    return out.astype(np.float64), R


def b_theta_phi_m(off_along, off_athwart, bw_along, bw_athwart):
    """BB transceiver gain compensation.  (calibrate_ek.py:507-530)"""
    with np.errstate(invalid="ignore", divide="ignore"):
        fa = (np.abs(-np.asarray(off_along, float)) / (np.asarray(bw_along, float) / 2)) ** 2
        ft = (np.abs(-np.asarray(off_athwart, float)) / (np.asarray(bw_athwart, float) / 2)) ** 2
        B = 0.5 * 6.0206 * (fa + ft - 0.18 * fa * ft)
    return np.where(np.isnan(B), 0.0, B)


def cal_complex_ek80(
    backscatter_r,
    backscatter_i,
    *,
    waveform_mode,
    cal_type,
    sample_interval,
    sound_speed,
    absorption,
    transmit_power,
    tau_nominal,
    gain,
    sa_correction,
    psi_fc,
    f_center,
    tau_eff,
    z_er,
    z_et,
    replicas=None,
    gpt=None,
):
    """Complex-sample Sv / TS for EK80 CW and BB.  (calibrate_ek.py:532-659)

    ``gain`` must already include the BB compensation (gain - B_theta_phi_m, :561-562).
    ``replicas`` (list of C arrays) is required for BB.
    Returns (out, echo_range, prx).
    """
    C, P, S, B = backscatter_r.shape
    cw = _cp(sound_speed, C, P)
    R = range_ek(backscatter_r, sample_interval, sound_speed)
    wavelength = cw / _cp(f_center, C, P)  # :568
    Rt = tvg_range_ek(R, "EK80", sample_interval, sound_speed, tau_nominal, gpt)
    x = backscatter_r + 1j * backscatter_i
    with np.errstate(invalid="ignore", divide="ignore"):
        Rt = np.where(Rt > 0, Rt, np.nan)  # :575
        spreading = 20 * np.log10(Rt)
        absorb = 2 * _cp(absorption, C, P) * Rt
        prx = ek80.power_from_complex(
            x, _cp(z_er, C, P), _cp(z_et, C, P), replicas if waveform_mode == "BB" else None
        )
        prx = np.where(prx > 0, prx, np.nan)  # :581
        Pt = _cp(transmit_power, C, P)
        g = _cp(gain, C, P)
        if cal_type == "Sv":
            out = (
                10 * np.log10(prx)
                + spreading
                + absorb
                - 10 * np.log10(wavelength**2 * Pt * cw / (32 * np.pi**2))
                - 2 * g
                - 10 * np.log10(_cp(tau_eff, C, P))
                - _cp(psi_fc, C, P)
            )  # :613-621
            if waveform_mode == "CW":
                out = out - 2 * _cp(sa_correction, C, P)  # :624-625
        elif cal_type == "TS":
            out = (
                10 * np.log10(prx)
                + 2 * spreading
                + absorb
                - 10 * np.log10(wavelength**2 * Pt / (16 * np.pi**2))
                - 2 * g
            )  # :630-637
        else:
            raise ValueError(cal_type)
    return out, R, prx


def range_azfp(S, *, cal_type, sound_speed, tau, n_avg, dig_rate, lockout, C, P):
    """AZFP echo_range.  (range.py:69-95)

    R = c*L/(2f) + (c/4) * (((2(s+1)-1)*N*1 - 1)/f + tau) - offset, offset = c*tau/4 for TS.
    """
    c = _cp(sound_speed, C, P)
    tau = _cp(tau, C, P)
    N, f, L = _cp(n_avg, C, P), _cp(dig_rate, C, P), _cp(lockout, C, P)
    s = np.arange(S, dtype=np.int64)[None, None, :]
    offset = 0 if cal_type == "Sv" else c * tau / 4
    return c * L / (2 * f) + (c / 4) * (((2 * (s + 1) - 1) * N * 1 - 1) / f + tau) - offset


def cal_azfp(
    counts,
    *,
    cal_type,
    sound_speed,
    absorption,
    tau,
    n_avg,
    dig_rate,
    lockout,
    EL,
    DS,
    TVR,
    VTX0,
    psi_lin,
    Sv_offset,
):
    """AZFP Sv / TS.  (calibrate_azfp.py:49-111)  Returns (out, echo_range)."""
    C, P, S = counts.shape
    R = range_azfp(
        S,
        cal_type=cal_type,
        sound_speed=sound_speed,
        tau=tau,
        n_avg=n_avg,
        dig_rate=dig_rate,
        lockout=lockout,
        C=C,
        P=P,
    )
    R = np.broadcast_to(R, (C, P, S))
    with np.errstate(invalid="ignore", divide="ignore"):
        spreading = 20 * np.log10(R)  # :64
        absorb = 2 * _cp(absorption, C, P) * R  # :65
        SL = _cp(TVR, C, P) + 20 * np.log10(_cp(VTX0, C, P))  # :66
        a = _cp(DS, C, P)
        ELv = _cp(EL, C, P) - 2.5 / a + counts / (26214 * a)  # :70-74
        if cal_type == "Sv":
            out = (
                ELv
                - SL
                + spreading
                + absorb
                - 10 * np.log10(0.5 * _cp(sound_speed, C, P) * _cp(tau, C, P) * _cp(psi_lin, C, P))
                + _cp(Sv_offset, C, P)
            )  # :78-91
        elif cal_type == "TS":
            out = ELv - SL + 2 * spreading + absorb  # :96
        else:
            raise ValueError("cal_type not recognized!")
    return out.astype(np.float64), np.array(R, dtype=np.float64)
