"""Oracle: EK80 transmit replica, pulse compression and received power (test infrastructure).

Restates /root/reference/echopype/calibrate/ek80_complex.py and the complex-sample
power computation of calibrate_ek.py:456-505.  The leaf functions here are pinned
against the reference's own outputs (tests/golden/ref_leaf_goldens.npz, produced by
oracle/gen_goldens.py with the stub-module loader of SURVEY Appendix B).

Third-party arithmetic: scipy.signal.convolve (scipy unpinned in the reference's
requirements.txt:17; 1.15.3 in this image), used exactly where the reference uses it
(ek80_complex.py:70,74,187,310) with method left at "auto".
"""
import numpy as np
from scipy import signal

__all__ = [
    "chirp_replica",
    "filter_and_decimate",
    "transmit_replica",
    "tau_effective",
    "norm_factor",
    "compress_pulse",
    "power_from_complex",
]


def chirp_replica(fs, tau, slope, f0, f1, drop_last_hanning_zero=False):
    """Hann-tapered linear chirp, peak-normalised.  (ek80_complex.py:12-52)

    All arguments scalar.  n = floor(tau * float32(fs)) (:30: the float32 cast of
    fs is in the reference); taper length L = round(2*tau*fs*slope) (:35) split in
    two halves applied to both ends (:38-50).
    """
    n = int(np.floor(tau * np.float32(fs)))
    t = np.linspace(0, n - 1, num=n) * 1 / fs
    sweep = np.pi * (f1 - f0) / tau
    y = np.cos(sweep * t * t + 2 * np.pi * f0 * t)
    L = int(np.round(tau * fs * slope * 2.0))
    win = 0.5 * (1.0 - np.cos(2.0 * np.pi * np.arange(0, L, 1) / (L - 1)))
    head = win[: L // 2]
    tail = win[L // 2 : -1] if drop_last_hanning_zero else win[L // 2 :]
    y[: head.size] *= head
    y[n - tail.size :] *= tail
    return y / np.max(y), t


def filter_and_decimate(y, wbt_fil, wbt_deci, pc_fil, pc_deci, fs):
    """WBT filter -> decimate -> PC filter -> decimate.  (ek80_complex.py:55-80)"""
    stage1 = signal.convolve(y, wbt_fil)[0 :: int(wbt_deci)]
    stage2 = signal.convolve(stage1, pc_fil)[0 :: int(pc_deci)]
    t = np.arange(stage2.size) * 1 / fs * wbt_deci * pc_deci
    return stage2, t


def transmit_replica(fs, tau, slope, f0, f1, filt, drop_last_hanning_zero=False):
    """Per-channel transmit replica as built by get_transmit_signal (ek80_complex.py:211-282).

    ``filt`` = dict(wbt_fil, wbt_decifac, pc_fil, pc_decifac) with NaN padding already
    dropped (get_vend_filter_EK80 :122-123).  For CW the caller passes f0 = f1 = f_nominal
    (:256-261).
    """
    y, _ = chirp_replica(fs, tau, slope, f0, f1, drop_last_hanning_zero)
    return filter_and_decimate(
        y, filt["wbt_fil"], filt["wbt_decifac"], filt["pc_fil"], filt["pc_decifac"], fs
    )


def tau_effective(ytx, fs_deci, waveform_mode):
    """Effective pulse length of one replica.  (ek80_complex.py:183-190)"""
    if waveform_mode == "BB":
        acorr = signal.convolve(ytx, np.flip(np.conj(ytx))) / np.linalg.norm(ytx) ** 2
        p = np.abs(acorr) ** 2
    elif waveform_mode == "CW":
        p = np.abs(ytx) ** 2
    else:
        raise ValueError(waveform_mode)
    return p.sum() / (p.max() * fs_deci)


def norm_factor(ytx):
    """||tx||^2  (ek80_complex.py:386-390)"""
    return np.linalg.norm(ytx) ** 2


def compress_pulse(x, replicas):
    """Matched filter along range_sample.  (ek80_complex.py:285-369)

    x        : complex (C, P, S, B), NaN-padded
    replicas : list of C complex 1-D transmit replicas

    Mirrors the reference: NaN -> 0 (:339-340); for every (ping, beam) slab
    [np.vectorize loop of xr.apply_ufunc, :356-364] an all-zero slab across channels
    is returned unchanged (:300-301), otherwise each channel is convolved with
    flipud(conj(tx)) in "full" mode and cropped at m-1 (:310-312); the result is
    stored as complex64 (:304); NaN restored where the input was NaN (:367).
    """
    C, P, S, B = x.shape
    nan_mask = np.isnan(x)
    xz = np.where(nan_mask, 0.0 + 0.0j, x)
    flipped = [np.flipud(np.conj(r)) for r in replicas]
    out = np.zeros((C, P, S, B), dtype=np.complex64)
    for p in range(P):
        for b in range(B):
            slab = xz[:, p, :, b]  # (C, S)
            if np.all(slab == 0.0 + 0.0j):
                continue  # zeros stay zeros
            for c in range(C):
                m = flipped[c].size
                out[c, p, :, b] = signal.convolve(slab[c], flipped[c], mode="full")[m - 1 :]
    # xr.where(nan_mask, nan, pc): complex NaN where the input was NaN
    out = np.where(nan_mask, np.complex64(np.nan), out)
    return out


def _nanmean_beam(z):
    """xarray .mean(dim="beam") on a complex array: NaN-skipping (skipna default True
    for float/complex dtypes).  All-NaN -> NaN."""
    valid = ~np.isnan(z)
    n = valid.sum(axis=-1)
    s = np.where(valid, z, 0).sum(axis=-1)
    with np.errstate(invalid="ignore", divide="ignore"):
        return np.where(n > 0, s / np.where(n > 0, n, 1), np.nan + 0j)


def power_from_complex(x, z_er, z_et, replicas=None):
    """Received power from complex samples.  (calibrate_ek.py:456-505)

    x : complex (C, P, S, B); z_er, z_et broadcastable to (C, P, 1) / (C,1,1).
    BB when ``replicas`` given: pulse-compress and divide by ||tx||^2 (:493-497).
    prx = B * |mean_beam|^2 / (2 sqrt 2)^2 * (|z_er + z_et| / z_er)^2 / z_et  (:483-490)
    """
    B = x.shape[-1]
    if replicas is not None:
        pc = compress_pulse(x, replicas)
        nf = np.array([norm_factor(r) for r in replicas])
        sig = pc / nf[:, None, None, None]  # complex64 / float64 -> complex128
    else:
        sig = x
    z_er = np.asarray(z_er, dtype=np.float64)
    z_et = np.asarray(z_et, dtype=np.float64)
    mean_b = _nanmean_beam(sig)
    return B * np.abs(mean_b) ** 2 / (2 * np.sqrt(2)) ** 2 * (np.abs(z_er + z_et) / z_er) ** 2 / z_et
