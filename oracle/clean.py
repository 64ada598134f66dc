"""Oracle: De Robertis & Higginbottom background-noise estimate/removal (test infrastructure).

Restates /root/reference/echopype/clean/api.py:362-433 (estimate_background_noise) and
:436-511 (remove_background_noise), with the dB-string parser of clean/utils.py:13-26.
Pinned by the reference's own synthetic known-answer test
(/root/reference/echopype/tests/clean/test_noise.py:902-987, restated in
tests/test_oracle_kat.py: NaN at samples 30/60; seed(1) => exactly 6 NaNs).
"""
import re
import warnings

import numpy as np

__all__ = ["extract_dB", "coarsen_mean", "estimate_background_noise", "remove_background_noise"]


def extract_dB(s):
    """'3.0dB' -> 3.0  (clean/utils.py:13-26, same error types/messages)."""
    if not isinstance(s, str):
        raise TypeError(
            "Decibal input must be a string formatted as `NUMdB` or `NUMdb."
            f"Cannot be of type `{type(s)}`."
        )
    m = re.search(r"^[-+]?\d+\.?\d*(?:dB|db)$", s, flags=re.IGNORECASE)
    if not m:
        raise ValueError("Decibal string must be formatted as 'NUMdB' or `NUMdb")
    return float(m.group(0)[:-2])


def coarsen_mean(a, ping_num, range_sample_num, skipna=True, func="mean"):
    """xarray ``.coarsen(ping_time=N, range_sample=M, boundary="pad").mean()/.min()``
    on a (C,P,S) array: tail blocks are NaN-padded, reduction is NaN-skipping
    (xarray default for float dtypes), all-NaN block -> NaN."""
    C, P, S = a.shape
    Pb, Sb = -(-P // ping_num), -(-S // range_sample_num)
    pad = np.full((C, Pb * ping_num, Sb * range_sample_num), np.nan)
    pad[:, :P, :S] = a
    blk = pad.reshape(C, Pb, ping_num, Sb, range_sample_num)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        if func == "mean":
            return np.nanmean(blk, axis=(2, 4)) if skipna else np.mean(blk, axis=(2, 4))
        if func == "min":
            return np.nanmin(blk, axis=(2, 4)) if skipna else np.min(blk, axis=(2, 4))
    raise ValueError(func)


def _alpha(sound_absorption, C, P):
    a = np.asarray(sound_absorption, dtype=np.float64)
    if a.ndim == 0:
        return a
    if a.ndim == 1:
        return a[:, None, None]
    return a[:, :, None]


def estimate_background_noise(
    Sv, echo_range, sound_absorption, ping_num, range_sample_num, background_noise_max=None
):
    """Sv_noise (C,P,S).  (clean/api.py:392-431)"""
    C, P, S = Sv.shape
    if background_noise_max is not None:
        background_noise_max = extract_dB(background_noise_max)  # :392-394
    with np.errstate(invalid="ignore", divide="ignore"), warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        # :397 -- where(echo_range >= 1, other=1) also maps NaN ranges to 1
        spreading = 20 * np.log10(np.where(echo_range >= 1, echo_range, 1))
        absorb = 2 * _alpha(sound_absorption, C, P) * echo_range  # :398
        power_cal = 10 ** ((Sv - spreading - absorb) / 10)  # :401
        binned = 10 * np.log10(coarsen_mean(power_cal, ping_num, range_sample_num))  # :402-408
        noise = np.nanmin(binned, axis=2)  # :411  (C, Pb); all-NaN -> NaN
        if background_noise_max is not None:  # :418-422
            noise = np.where(noise < background_noise_max, noise, background_noise_max)
        # :425-428 forward-fill to every ping of the block
        up = noise[:, np.arange(P) // ping_num]
        return up[:, :, None] + spreading + absorb  # :429-430


def remove_background_noise(
    Sv,
    echo_range,
    sound_absorption,
    ping_num,
    range_sample_num,
    background_noise_max=None,
    SNR_threshold="3.0dB",
):
    """(Sv_noise, Sv_corrected).  (clean/api.py:472-487)"""
    snr = extract_dB(SNR_threshold) if SNR_threshold is not None else None
    Sv_noise = estimate_background_noise(
        Sv, echo_range, sound_absorption, ping_num, range_sample_num, background_noise_max
    )
    with np.errstate(invalid="ignore", divide="ignore"):
        lin = 10 ** (Sv / 10) - 10 ** (Sv_noise / 10)  # :485
        corr = 10 * np.log10(np.where(lin > 0, lin, np.nan))  # :486
        corr = np.where(corr - Sv_noise > snr, corr, np.nan)  # :487
    return Sv_noise, corr
