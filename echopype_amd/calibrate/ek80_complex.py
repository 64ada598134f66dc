"""EK80 transmit replica (host side; a few hundred samples per channel).

Builds what the matched-filter kernel needs from the Vendor_specific filter coefficients and the
per-channel transmit parameters, following /root/reference/echopype/calibrate/ek80_complex.py:
tapered_chirp :12-52, filter_decimate_chirp :55-80, get_vend_filter_EK80 / get_filter_coeff
:83-159, get_tau_effective :162-208, get_transmit_signal :211-282, get_norm_fac :372-391.
The sample-sized work (compress_pulse :285-369) runs on the GPU (csrc/ek80_complex.hip).
"""
from collections import defaultdict

import numpy as np
from scipy import signal

from ..xr_lite import DataArray

__all__ = ["tapered_chirp", "filter_decimate_chirp", "get_vend_filter_EK80", "get_filter_coeff",
           "get_tau_effective", "get_transmit_signal", "get_norm_fac"]


def _scalar(x):
    return float(np.asarray(x).reshape(-1)[0])


def tapered_chirp(fs, transmit_duration_nominal, slope, transmit_frequency_start, transmit_frequency_stop,
                  drop_last_hanning_zero=False):
    """Hann-tapered linear FM pulse normalised to unit peak; returns (y, t)."""
    fs, tau = _scalar(fs), _scalar(transmit_duration_nominal)
    f0, f1, slope = _scalar(transmit_frequency_start), _scalar(transmit_frequency_stop), _scalar(slope)
    n = int(np.floor(tau * np.float32(fs)))          # sample count uses float32(fs), as the vendor code
    t = np.linspace(0, n - 1, num=n) * 1 / fs
    phase = (np.pi * (f1 - f0) / tau) * t * t + (2 * np.pi * f0) * t
    y = np.cos(phase)
    L = int(np.round(tau * fs * slope * 2.0))
    hann = 0.5 * (1.0 - np.cos(2.0 * np.pi * np.arange(0, L, 1) / (L - 1)))
    half = int(len(hann) / 2)
    rise, fall = hann[:half], (hann[half:-1] if drop_last_hanning_zero else hann[half:])
    y[: len(rise)] = y[: len(rise)] * rise
    y[n - len(fall):] = y[n - len(fall):] * fall
    return y / np.max(y), t


def filter_decimate_chirp(coeff_ch, y_ch, fs):
    """Two filter+decimate stages (WBT then PC) applied to the ideal pulse."""
    fs = _scalar(fs)
    y = y_ch
    for fil, dec in ((coeff_ch["wbt_fil"], coeff_ch["wbt_decifac"]), (coeff_ch["pc_fil"], coeff_ch["pc_decifac"])):
        y = signal.convolve(y, fil)[0:: int(dec)]
    t = np.arange(y.size) * 1 / fs * coeff_ch["wbt_decifac"] * coeff_ch["pc_decifac"]
    return y, t


def get_vend_filter_EK80(vend, channel_id, filter_name, param_type):
    """Complex filter taps (NaN padding dropped) or decimation factor of one channel; None if absent."""
    names = [f"{filter_name}_coeffs_imag", f"{filter_name}_coeffs_real", f"{filter_name}_deci_fac"]
    if not all(n in vend for n in names):
        return None
    i = list(map(str, vend["channel"].values)).index(str(channel_id))
    if param_type == "coeff":
        c = np.asarray(vend[names[1]].values)[i] + 1j * np.asarray(vend[names[0]].values)[i]
        return c[~np.isnan(c)]
    return np.asarray(vend[names[2]].values)[i]


def get_filter_coeff(vend):
    if "filter_time" in vend.sizes:
        vend = vend.isel(filter_time=0)
    coeff = defaultdict(dict)
    for ch in vend["channel"].values:
        coeff[ch]["wbt_fil"] = get_vend_filter_EK80(vend, ch, "WBT", "coeff")
        coeff[ch]["pc_fil"] = get_vend_filter_EK80(vend, ch, "PC", "coeff")
        coeff[ch]["wbt_decifac"] = get_vend_filter_EK80(vend, ch, "WBT", "decimation")
        coeff[ch]["pc_decifac"] = get_vend_filter_EK80(vend, ch, "PC", "decimation")
    return coeff


def get_tau_effective(ytx_dict, fs_deci_dict, waveform_mode, channel=None, ping_time=None):
    """Effective pulse length per channel -> DataArray(channel)."""
    vals = []
    for ch, ytx in ytx_dict.items():
        if waveform_mode == "BB":
            a = signal.convolve(ytx, np.flip(np.conj(ytx))) / np.linalg.norm(ytx) ** 2
            p = np.abs(a) ** 2
        elif waveform_mode == "CW":
            p = np.abs(ytx) ** 2
        else:
            raise ValueError(waveform_mode)
        vals.append(_scalar(p.sum() / (p.max() * fs_deci_dict[ch])))
    ch_vals = np.asarray(list(ytx_dict)) if channel is None else np.asarray(getattr(channel, "values", channel))
    return DataArray(np.asarray(vals, dtype=np.float64), ("channel",), {"channel": ch_vals})


def get_transmit_signal(beam, coeff, waveform_mode, fs, drop_last_hanning_zero=False, *, whole_file=None):
    """Per-channel replica and its time axis; transmit parameters must be constant across pings.
    ``whole_file`` (a ping shard; sharding.file_scalars): {parameter: (min (C,), max (C,))} over the pings of the WHOLE
    file (or filter interval) -- the uniqueness test and the value are the file's, not the shard's, so that a shard
    holding none (or only some) of a channel's valid pings builds the same replica and raises the same error."""
    tt = np.asarray(beam["transmit_type"].values) if "transmit_type" in beam else None
    if waveform_mode == "BB" and tt is not None and tt.size and np.all(tt == "CW"):  # (tt empty: a shard without pings)
        raise TypeError("File does not contain BB mode complex samples!")
    chans = list(beam["channel"].values)
    fs_all = np.asarray(getattr(fs, "values", fs), dtype=np.float64)
    y_all, t_all = {}, {}
    for i, ch in enumerate(chans):
        prm = {}
        for p in ("transmit_duration_nominal", "slope", "transmit_frequency_start", "transmit_frequency_stop"):
            if waveform_mode == "CW" and p.startswith("transmit_frequency"):
                v = np.unique(np.asarray(beam["frequency_nominal"].values)[i])
            elif whole_file is not None and p in whole_file:
                lo, hi = float(whole_file[p][0][i]), float(whole_file[p][1][i])
                v = np.array([lo]) if (lo == hi and np.isfinite(lo)) else (np.array([]) if lo > hi else np.array([lo, hi]))
            else:
                v = np.unique(np.asarray(beam[p].values)[i])
                v = v[~np.isnan(v)]
            if v.size != 1:
                raise TypeError("File contains changing %s!" % p)
            prm[p] = v
        fs_ch = fs_all if fs_all.ndim == 0 else fs_all[i]
        y, _ = tapered_chirp(fs=fs_ch, drop_last_hanning_zero=drop_last_hanning_zero, **prm)
        y_all[ch], t_all[ch] = filter_decimate_chirp(coeff[ch], y, fs_ch)
    return y_all, t_all


def get_norm_fac(chirp):
    return DataArray(np.array([np.linalg.norm(tx) ** 2 for tx in chirp.values()]), ("channel",),
                     {"channel": np.asarray(list(chirp))})
