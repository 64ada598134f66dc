"""Oracle: Ryan et al. (2015) noise masks and apply_mask (test infrastructure).

Restates /root/reference/echopype/clean/api.py:30-359 (mask_transient_noise, mask_impulse_noise,
mask_attenuated_signal), clean/utils.py:29-377 (pooling / down-up-sampling helpers and the two
echopy single-channel leaf functions) and mask/api.py:307-464 (apply_mask, array part).

Pinning: the two pure-numpy leaf functions (echopy_impulse_noise_mask, echopy_attenuated_signal_mask)
are pinned against the reference's own code executed through the stub loader
(oracle/gen_goldens.py -> tests/golden/ref_mask_goldens.npz).  The xarray/flox/dask_image based
helpers cannot run here; they are restated from their definitions and pinned by restating the
reference's structural tests (tests/clean/test_noise.py:342-441 symmetric-pad window check,
:616-683 coarsen check) with scipy.ndimage.generic_filter (what dask_image wraps) as the engine.

Arrays are (channel, ping_time, range_sample) unless stated otherwise.
"""
import warnings

import numpy as np
import scipy.ndimage

from .clean import coarsen_mean, extract_dB
from .commongrid import parse_range_bin

__all__ = [
    "nsamples_per_bin", "index_binning_downsample_upsample", "downsample_upsample",
    "echopy_impulse_noise_mask", "mask_impulse_noise", "echopy_attenuated_signal_mask",
    "mask_attenuated_signal", "index_binning_pool_Sv", "pool_Sv", "mask_transient_noise",
    "apply_mask",
]


def _lin(x):
    with np.errstate(over="ignore", invalid="ignore"):
        return 10 ** (x / 10)  # utils/compute.py:14-27


def _log(x):
    with np.errstate(divide="ignore", invalid="ignore"):
        return 10 * np.log10(x)  # utils/compute.py:30-42


def nsamples_per_bin(range_arr, depth_bin):
    """Per-channel number of range samples covering ``depth_bin`` metres (clean/utils.py:129-133)."""
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        step = np.nanmean(np.diff(range_arr, axis=2), axis=(1, 2))
    return np.ceil(depth_bin / step).astype(int)


def index_binning_downsample_upsample(Sv, range_arr, depth_bin):
    """clean/utils.py:244-304: per channel coarsen(range_sample=n_c, boundary="pad").mean(skipna)
    in the linear domain, then forward-fill every block value over its n_c samples."""
    C, P, S = Sv.shape
    n = nsamples_per_bin(range_arr, depth_bin)
    up = np.empty((C, P, S))
    for c in range(C):
        coarse = _log(coarsen_mean(_lin(Sv[c:c + 1]), 1, int(n[c])))[0]  # (P, ceil(S/n))
        up[c] = coarse[:, np.arange(S) // int(n[c])]
    return up


def downsample_upsample(Sv, range_arr, depth_bin):
    """clean/utils.py:173-241: bins arange(min, max + bin, bin) closed on the left (flox nanmean
    per (channel, ping, bin) in the linear domain); every sample then takes the value of the bin
    np.digitize puts it in.  (The reference builds the up-sampling from the first sample of each bin
    and a forward fill, which is this whenever the range variable increases along range_sample and
    every bin is hit in every ping -- otherwise the reference raises on a coordinate-length mismatch.)"""
    C, P, S = Sv.shape
    dmin, dmax = np.nanmin(range_arr), np.nanmax(range_arr)
    edges = np.arange(dmin, dmax + depth_bin, depth_bin)
    nb = len(edges) - 1
    lin = _lin(Sv)
    down = np.full((C, P, nb), np.nan)
    up = np.full((C, P, S), np.nan)
    for c in range(C):
        for p in range(P):
            d = range_arr[c, p]
            b = np.searchsorted(edges, d, side="right") - 1  # [e_j, e_j+1)
            ok = (b >= 0) & (b < nb) & np.isfinite(d) & ~np.isnan(lin[c, p])
            s = np.bincount(b[ok], weights=lin[c, p][ok], minlength=nb)
            n = np.bincount(b[ok], minlength=nb)
            with np.errstate(invalid="ignore", divide="ignore"):
                down[c, p] = _log(s / n)
            dig = np.digitize(d, edges[:-1])  # NaN -> nb
            up[c, p] = down[c, p][np.clip(dig - 1, 0, nb - 1)]
    return down, up


def echopy_impulse_noise_mask(Sv, num_side_pings, impulse_noise_threshold):
    """(range_sample, ping_time) two-sided ping comparison (clean/utils.py:307-323)."""
    S, P = Sv.shape
    n = num_side_pings
    fwd = np.full((S, P), np.nan)
    bwd = np.full((S, P), np.nan)
    with np.errstate(invalid="ignore"):
        fwd[:, :max(P - n, 0)] = Sv[:, :max(P - n, 0)] - Sv[:, n:]
        bwd[:, n:] = Sv[:, n:] - Sv[:, :max(P - n, 0)]
    fwd[np.isnan(fwd)] = np.inf
    bwd[np.isnan(bwd)] = np.inf
    return (fwd > impulse_noise_threshold) & (bwd > impulse_noise_threshold)


def mask_impulse_noise(Sv, range_arr, depth_bin="5m", num_side_pings=2,
                       impulse_noise_threshold="10.0dB", use_index_binning=False):
    """clean/api.py:171-266.  Returns a (channel, range_sample, ping_time) boolean array (the
    reference's apply_ufunc moves its core dims [range_sample, ping_time] last)."""
    thr = extract_dB(impulse_noise_threshold)
    depth_bin = parse_range_bin(depth_bin)
    if use_index_binning:
        up = index_binning_downsample_upsample(Sv, range_arr, depth_bin)
    else:
        _, up = downsample_upsample(Sv, range_arr, depth_bin)
    return np.stack([echopy_impulse_noise_mask(up[c].T, num_side_pings, thr) for c in range(Sv.shape[0])])


def _argmin_like_numpy(a):
    return int(np.argmin(a))  # first NaN if any, as np.argmin does


def echopy_attenuated_signal_mask(Sv, range_var, upper_limit_sl, lower_limit_sl, num_side_pings,
                                  attenuation_signal_threshold):
    """(ping_time, range_sample) ping-vs-block median comparison (clean/utils.py:326-372)."""
    P, S = Sv.shape
    n = num_side_pings
    mask = np.zeros((P, S), dtype=bool)
    for p in range(P):
        up = _argmin_like_numpy(np.abs(range_var[p] - upper_limit_sl))
        lw = _argmin_like_numpy(np.abs(range_var[p] - lower_limit_sl))
        if p - n < 0 or p + n > P - 1 or np.all(np.isnan(Sv[p, up:lw])):
            continue
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", RuntimeWarning)
            pingmedian = _log(np.nanmedian(_lin(Sv[p, up:lw])))
            blockmedian = _log(np.nanmedian(_lin(Sv[p - n:p + n, up:lw])))
        if pingmedian - blockmedian < attenuation_signal_threshold:
            mask[p, :] = True
    return mask


def mask_attenuated_signal(Sv, range_arr, upper_limit_sl="400.0m", lower_limit_sl="500.0m",
                           num_side_pings=15, attenuation_signal_threshold="8.0dB"):
    """clean/api.py:269-359 -> (channel, ping_time, range_sample) boolean."""
    if upper_limit_sl > lower_limit_sl:  # :308 (string comparison in the reference, kept)
        raise ValueError("Minimum range has to be shorter than maximum range")
    thr = extract_dB(attenuation_signal_threshold)
    lw = parse_range_bin(lower_limit_sl)
    up = parse_range_bin(upper_limit_sl)
    if up > np.nanmax(range_arr) or lw < np.nanmin(range_arr):  # :322-324
        return np.zeros(Sv.shape, dtype=bool)
    return np.stack([echopy_attenuated_signal_mask(Sv[c], range_arr[c], up, lw, num_side_pings, thr)
                     for c in range(Sv.shape[0])])


def index_binning_pool_Sv(Sv, range_arr, func, depth_bin, num_side_pings, exclude_above):
    """clean/utils.py:109-170: (2n+1) x (2m_c+1) window aggregate (reflect boundary) of the linear Sv
    below ``exclude_above``; samples above it are NaN."""
    C, P, S = Sv.shape
    m = nsamples_per_bin(range_arr, depth_bin)
    with np.errstate(invalid="ignore"):
        # :143 flat argmin over the whole (C,P,S) array (first element that is NOT <= exclude_above)
        s0 = int(np.argmin(range_arr <= exclude_above))
    pooled = np.full((C, P, S), np.nan)
    if s0 >= S:
        return pooled
    for c in range(C):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", RuntimeWarning)
            win = scipy.ndimage.generic_filter(
                _lin(Sv[c, :, s0:]), func, size=[2 * num_side_pings + 1, 2 * int(m[c]) + 1], mode="reflect")
        pooled[c, :, s0:] = _log(win)
    return pooled


def pool_Sv(Sv, range_arr, func, depth_bin, num_side_pings, exclude_above):
    """clean/utils.py:29-106: value windows [d - bin, d + bin] x [p - n, p + n] (triple loop)."""
    C, P, S = Sv.shape
    n = num_side_pings
    dmin, dmax = np.nanmin(range_arr), np.nanmax(range_arr)
    pooled = np.full((C, P, S), np.nan)
    pidx = np.arange(P)[:, None]
    for c in range(C):
        lin = _lin(Sv[c])
        for s in range(S):
            for p in range(P):
                d = range_arr[c, p, s]
                if not (d - depth_bin >= dmin and d + depth_bin <= dmax and d - depth_bin >= exclude_above
                        and p - n >= 0 and p + n <= P):
                    continue
                with np.errstate(invalid="ignore"):
                    w = ((d - depth_bin <= range_arr[c]) & (range_arr[c] <= d + depth_bin)
                         & (p - n <= pidx) & (pidx <= p + n))
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore", RuntimeWarning)
                    pooled[c, p, s] = _log(func(np.where(w, lin, np.nan)))
    return pooled


def mask_transient_noise(Sv, range_arr, func="nanmean", depth_bin="10m", num_side_pings=25,
                         exclude_above="250.0m", transient_noise_threshold="12.0dB",
                         use_index_binning=False):
    """clean/api.py:30-168 -> (channel, ping_time, range_sample) boolean."""
    if func not in ("nanmean", "nanmedian"):
        raise ValueError(f"Input `func` is `{func}`. `func` must be `nanmean` or `nanmedian`.")
    f = np.nanmean if func == "nanmean" else np.nanmedian
    thr = extract_dB(transient_noise_threshold)
    depth_bin = parse_range_bin(depth_bin)
    exclude_above = parse_range_bin(exclude_above)
    pool = index_binning_pool_Sv if use_index_binning else pool_Sv
    pooled = pool(Sv, range_arr, f, depth_bin, num_side_pings, exclude_above)
    with np.errstate(invalid="ignore"):
        return Sv - pooled > thr


def apply_mask(src, masks, fill_value=np.nan):
    """mask/api.py:402-432 array part: logical AND of the (broadcast) masks, NaN -> False, then
    where(mask, src, fill_value).  ``src`` (C,P,S); each mask (C,P,S) or (P,S)."""
    if not isinstance(masks, (list, tuple)):
        masks = [masks]
    final = np.ones(src.shape, dtype=bool)
    for m in masks:
        m = np.asarray(m)
        m = np.where(np.isnan(m.astype(float)), False, m).astype(bool)
        final &= np.broadcast_to(m, src.shape)
    return np.where(final, src, fill_value)
