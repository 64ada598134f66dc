"""estimate_background_noise / remove_background_noise with the reference's signatures
(/root/reference/echopype/clean/api.py:362-433, :436-511; De Robertis & Higginbottom 2007).
The per-sample work is two HIP kernel launches (epa_noise_estimate, epa_noise_apply).
``remove_background_noise`` adds ``Sv_noise`` and ``Sv_corrected`` to the CALLER's dataset, as the
reference does (api.py:490-502).

Also the Ryan et al. (2015) noise masks (api.py:30-359: mask_transient_noise, mask_impulse_noise,
mask_attenuated_signal), each a handful of launches from csrc/noise_masks.hip.
"""
import logging
import os
import weakref

import numpy as np
import torch

from .. import ops
from ..commongrid.api import _coef_rows, _dev, _full, _range_stats
from ..commongrid.utils import _parse_x_bin
from ..utils.prov import echopype_prov_attrs, insert_processing_level
from ..xr_lite import DataArray, DeviceArray, LazyDeviceArray, from_xarray, xarray_io
from .utils import add_remove_background_noise_attrs, extract_dB


def _inputs(ds_Sv):
    sv_da = ds_Sv["Sv"]
    order = tuple(sv_da.dims)
    sv_t = _dev(sv_da)
    if sv_t.dtype not in (torch.float32, torch.float64):
        sv_t = sv_t.double()
    # a lazy echo_range straight from compute_Sv on power samples: the kernels evaluate its coefficient rows and take the
    # NaN pattern from the raw samples (rg_t = (rows, raw)); the array is never written
    rows = _coef_rows(ds_Sv["echo_range"], order, sv_t)
    raw = ds_Sv["echo_range"].data.nan_source() if rows is not None else None
    if raw is not None and raw.dtype == torch.float32 and tuple(raw.shape) == tuple(sv_t.shape) and raw.is_contiguous():
        rg_t = (rows, raw)
    else:
        rg_t = _dev(_full(ds_Sv["echo_range"], ds_Sv, order), sv_t.dtype)
    C, P, S = sv_t.shape
    return order, sv_t, rg_t, _alpha2(ds_Sv, order, C, P)


def _alpha2(ds_Sv, order, C, P):
    """2 * sound_absorption of the dataset as a (C, P) float64 device tensor (clean/api.py:397-398 uses the dataset's)."""
    a = ds_Sv["sound_absorption"]
    av = np.asarray(a.values, dtype=np.float64)
    if av.ndim == 0:
        a2 = np.full((C, P), 2 * float(av))
    elif av.ndim == 1:
        a2 = np.broadcast_to((2 * av)[:, None] if a.dims == (order[0],) or av.shape[0] == C else (2 * av)[None, :], (C, P))
    else:
        a2 = 2 * (av if a.dims[0] == order[0] else av.T)
    return ops.to_device(np.ascontiguousarray(a2, dtype=np.float64))


class DenoiseSource:
    """What the ``Sv_noise`` / ``Sv_corrected`` that ``remove_background_noise`` left deferred are made from
    (``LazyDeviceArray.source``).  Pass 1 has run (the Sv array, the noise estimate); pass 2 -- raw -> Sv_noise,
    Sv_corrected and their minima / maxima (the ``actual_range`` attributes) -- runs when somebody reads one of the
    two arrays or attributes, or, the usual sequence, INSIDE ``compute_MVBS``'s pass when the corrected Sv is what is
    binned (``commongrid.api._mvbs_of_deferred_clean``: one sweep writes both arrays and the bins)."""

    def __init__(self, power, a2, noise, ping_num, snr, sv_t, ping_phase=0, global_rmax=None):
        self.power, self.a2, self.noise, self.ping_num, self.snr, self.sv_t = power, a2, noise, ping_num, snr, sv_t
        # a ping shard of a longer file: local ping p belongs to noise block (p + ping_phase) // ping_num, and
        # nanmax(echo_range) over ALL shards (a HostFuture; all-reduced in HBM behind pass 1) sizes the range grid
        self.ping_phase, self.global_rmax = int(ping_phase), global_rmax
        self.raw_version = power.raw._version
        self.minmax = None  # HostFuture / list of [min, max of Sv_noise, min, max of Sv_corrected] once pass 2 has run
        self._lazy = {}     # weak: the arrays own this object (source, make), not the other way round -- a dataset
        #                     dropped unread frees the Sv of pass 1 with it, no reference cycle to wait for

    def arrays(self):
        """The two deferred arrays (created once; the caller keeps them alive by putting them into its dataset)."""
        shape, dev = tuple(self.power.raw.shape), self.power.raw.device
        out = {k: LazyDeviceArray(shape, self.power.dtype, dev, (lambda k=k: self._make(k)), source=self)
               for k in ("noise", "corrected")}
        self._lazy = {k: weakref.ref(v) for k, v in out.items()}
        return out

    def lazy(self, which):
        ref = self._lazy.get(which)
        return ref() if ref is not None else None

    def intact(self):
        """The raw samples have not been written to since pass 1 looked at them."""
        return self.power.raw._version == self.raw_version

    def install(self, res):
        """The by-products of a pass 2 somebody ran (``ops.sv_denoise_mvbs`` with minmax_async)."""
        for k, name in (("noise", "Sv_noise"), ("corrected", "Sv_corrected")):
            la = self.lazy(k)
            if la is not None and not la.materialized:
                la.fulfil(res[name])
        self.minmax = res["minmax"]
        self.sv_t = None

    def run_plain(self):
        """Pass 2 alone: the chain kernel on plain 20-ping groups with ONE range bin (its bins are not wanted)."""
        from .. import _lib

        if not self.intact():
            raise RuntimeError("the raw samples were modified between remove_background_noise and the first read of "
                               "its deferred outputs (EPA_DEFER_CLEAN=0 writes them inside the call)")
        p = self.power
        P = p.raw.shape[1]
        group = 20
        bin_start = torch.arange(0, P + group, group, dtype=torch.int32, device=p.raw.device).clamp_(max=P)
        try:
            res = ops.sv_denoise_mvbs(p.raw, p.coef, self.a2, self.noise, self.ping_num, float(self.snr), bin_start,
                                      bin_start.numel() - 1, 1e30, 1, flags=p.flags, dtype=p.dtype, want_noise=True,
                                      want_corrected=True, want_minmax=True, minmax_async=True,
                                      ping_phase=self.ping_phase)
        except _lib.EpaError:  # a geometry the chain kernel does not serve: the array kernel on the Sv of pass 1
            sn, sc, mm = ops.noise_apply(self.sv_t, self.a2, self.noise, self.ping_num, float(self.snr), want_minmax=True,
                                         coef=p.coef, mask_raw=p.raw, ping_phase=self.ping_phase)
            res = dict(Sv_noise=sn, Sv_corrected=sc, minmax=mm)
        self.install(res)

    def _make(self, which):
        if self.minmax is None:
            self.run_plain()
        return self.lazy(which)._tensor

    def actual_range(self, which):
        if self.minmax is None:
            self.run_plain()
        mm = self.minmax.tolist() if hasattr(self.minmax, "tolist") else list(self.minmax)
        return mm[0:2] if which == "noise" else mm[2:4]


def defer_clean_enabled():
    """EPA_DEFER_CLEAN=0: remove_background_noise writes Sv_noise / Sv_corrected before it returns."""
    return os.environ.get("EPA_DEFER_CLEAN", "1") != "0"


def _denoise_deferred_sv(ds_Sv, ping_num, range_sample_num, nmax, snr, shard=None):
    """``remove_background_noise`` as the FIRST reader of the Sv that compute_Sv deferred (power samples): two passes over
    the raw samples instead of K1 + two sweeps of the Sv array --
        pass 1 (epa_sv_noise_fused)    raw -> the Sv array + the noise estimate from the values in registers + the
                                       echo_range statistics                                   4 + 8 B/sample
        pass 2 (epa_sv_denoise_mvbs)   raw -> Sv_noise, Sv_corrected, their actual_range      4 + 16 B/sample
    against 12 (K1) + 8 (estimate) + 24 (apply).  Pass 1 runs here; pass 2 is left to a ``DenoiseSource`` (see there).
    Nothing waits for the GPU.  Returns the DenoiseSource, or None: the plain route then runs on whatever this one
    has written.

    ``shard`` = (ping_offset, group, ShardContext): ``ds_Sv`` is one rank's ping shard.  Noise blocks count from the
    file's first ping (``ping_phase``); pass 1 also leaves the raw (sum, count) rows of the shard's first / last block,
    and a block cut by a shard edge gets the mean over ALL its pings (clean/api.py:402-411) through one all-reduce in
    HBM (sharding.merge_noise_edges) before pass 2 reads the noise.  Whether the route is taken must not depend on the
    rank (the collectives that follow differ): it is put to ONE vote -- a rank that cannot take it for a rank-local reason
    (its Sv already read, a kernel that declines) votes against, and then every rank takes the plain route."""
    from .. import _lib

    sv_da, rng_da = ds_Sv["Sv"], ds_Sv["echo_range"] if "echo_range" in ds_Sv else None
    d = sv_da.data
    src = d.source if isinstance(d, LazyDeviceArray) and not d.materialized else None
    dims = ("channel", "ping_time", "range_sample")
    ctx = None
    if shard is not None:
        from .. import sharding

        ctx = shard[2] if shard[2] is not None else sharding.ShardContext(shard[1])
    if src is None or isinstance(src, DenoiseSource) or rng_da is None or src.cal_type != "Sv" \
            or rng_da.data is not src.echo_range \
            or tuple(sv_da.dims) != dims or tuple(rng_da.dims) != dims \
            or src.echo_range.coef_rows() is not src.coef or not src.intact():
        # (rank-local state -- this rank's Sv already read, its raw samples written to: the other ranks of the file may be
        #  on their way into the vote below, and this one must cast its own before it takes the plain route)
        if ctx is not None:
            ctx.agree(True)
        return None
    C, P, S = d.shape
    a2 = _alpha2(ds_Sv, dims, C, P)
    phase = 0 if shard is None else int(shard[0]) % int(ping_num)
    out = None
    try:
        out = ops.sv_noise_fused(src.raw, src.coef, a2, ping_num, range_sample_num, flags=src.flags, dtype=src.dtype,
                                 noise_max=float("nan") if nmax is None else float(nmax), want_range_stats=True,
                                 ping_phase=phase, want_edges=shard is not None)
    except _lib.EpaError:  # e.g. more range blocks than the LDS holds
        pass
    # served by the generic kernel?  It leaves no range statistics (K1 does, on the plain route).  Known on the host
    # (epa_last_range_stats_filled): nothing waits for pass 1
    declined = out is None or not _lib.lib.epa_last_range_stats_filled()
    if shard is not None:
        declined = ctx.agree(declined)  # (every rank takes the plain route if any has to)
    if declined:
        return None
    sv_t, _, noise, rstats = out[:4]
    global_rmax = None
    if shard is not None:
        sharding.merge_noise_edges(noise, out[4], out[5], int(shard[0]), P, ping_num,
                                   float("nan") if nmax is None else float(nmax), shard[1], shard=ctx)
        # nanmax(echo_range) over all shards, for the range grid compute_MVBS will need: all-reduced where it lies
        global_rmax = ops.fetch_async(sharding.global_max_device(rstats[1:2].clone(), shard[1]))
    d.fulfil(sv_t)
    # (the three numbers start their way to the host now, behind pass 1 only: compute_MVBS sizes its grid from them)
    src.echo_range.set_stats(ops.fetch_async(rstats))
    dsrc = DenoiseSource(src, a2, noise, ping_num, snr, sv_t, ping_phase=phase, global_rmax=global_rmax)
    arrays = dsrc.arrays()
    if not defer_clean_enabled():
        dsrc.run_plain()
    return dsrc, arrays


def _rng_kw(rg_t, apply=False):
    """Keyword arguments of ops.noise_estimate / noise_apply for a range given as an array or as (rows, raw)."""
    if isinstance(rg_t, tuple):
        return dict(coef=rg_t[0], mask_raw=rg_t[1]) if apply else dict(coef=rg_t[0])
    return dict(range=rg_t)


def _estimate(ds_Sv, ping_num, range_sample_num, background_noise_max, shard=None):
    if background_noise_max is not None:
        background_noise_max = extract_dB(background_noise_max)
    order, sv_t, rg_t, a2 = _inputs(ds_Sv)
    nmax = float("nan") if background_noise_max is None else float(background_noise_max)
    if shard is None:
        noise = ops.noise_estimate(sv_t, a2, ping_num, range_sample_num, noise_max=nmax, **_rng_kw(rg_t))
    else:  # one rank's ping shard: blocks count from the dataset's first ping; blocks cut by a shard edge are merged
        from .. import sharding

        ping_offset, group, ctx = shard
        noise, es, ec = ops.noise_estimate(sv_t, a2, ping_num, range_sample_num, noise_max=nmax, **_rng_kw(rg_t),
                                           ping_phase=ping_offset % ping_num, want_edges=True)
        sharding.merge_noise_edges(noise, es, ec, ping_offset, sv_t.shape[1], ping_num, nmax, group, shard=ctx)
    return order, sv_t, rg_t, a2, noise, background_noise_max


@xarray_io()
def estimate_background_noise(ds_Sv, ping_num, range_sample_num, background_noise_max=None):
    """Sv_noise (same shape as Sv): block-minimum noise + transmission loss (api.py:392-431)."""
    ds_Sv = from_xarray(ds_Sv)
    order, sv_t, rg_t, a2, noise, _ = _estimate(ds_Sv, ping_num, range_sample_num, background_noise_max)
    sn, _ = ops.noise_apply(sv_t, a2, noise, ping_num, 0.0, want_corrected=False, **_rng_kw(rg_t, apply=True))
    return DataArray(DeviceArray(sn), order, {d: ds_Sv[d].values for d in order if d in ds_Sv.coords},
                     name="Sv_noise")


@xarray_io(in_place=("Sv_noise", "Sv_corrected"))
def remove_background_noise(ds_Sv, ping_num, range_sample_num, background_noise_max=None,
                            SNR_threshold="3.0dB", *, _shard=None):
    """Adds Sv_noise and Sv_corrected to ``ds_Sv`` and returns it (api.py:472-511).
    (``_shard`` = (ping_offset, group, ShardContext | None): set by echopype_amd.sharding.remove_background_noise.)"""
    ds_Sv = from_xarray(ds_Sv)
    if SNR_threshold is not None:
        SNR_threshold = extract_dB(SNR_threshold)
    done = None
    if SNR_threshold is not None:
        nmax = extract_dB(background_noise_max) if background_noise_max is not None else None
        done = _denoise_deferred_sv(ds_Sv, ping_num, range_sample_num, nmax, SNR_threshold, shard=_shard)
    if done is not None:
        order = tuple(ds_Sv["Sv"].dims)
        done, arrays = done
        sn, sc = arrays["noise"], arrays["corrected"]
        mm = None
    else:
        order, sv_t, rg_t, a2, noise, nmax = _estimate(ds_Sv, ping_num, range_sample_num, background_noise_max, _shard)
        # actual_range of both outputs is a by-product of the kernel (no extra sweeps)
        # (... read when somebody reads the attribute: the call does not wait for the kernel)
        sn, sc, mm = ops.noise_apply(sv_t, a2, noise, ping_num, float(SNR_threshold), want_minmax=True,
                                     minmax_async=defer_clean_enabled(),
                                     **_rng_kw(rg_t, apply=True), ping_phase=0 if _shard is None else _shard[0] % ping_num)
    for name, t, kind in (("Sv_noise", sn, "noise"), ("Sv_corrected", sc, "corrected")):
        da = DataArray(t if isinstance(t, DeviceArray) else DeviceArray(t), order)
        if done is not None:  # minimum / maximum: by-products of pass 2, read when somebody reads the attribute
            rng_mm = (lambda k=kind: done.actual_range(k))
        elif hasattr(mm, "tolist"):  # a HostFuture
            rng_mm = (lambda k=kind, f=mm: f.tolist()[0:2] if k == "noise" else f.tolist()[2:4])
        else:
            rng_mm = mm[0:2] if kind == "noise" else mm[2:4]
        ds_Sv[name] = add_remove_background_noise_attrs(da, kind, ping_num, range_sample_num, SNR_threshold,
                                                        nmax, rng_mm)
    prov = echopype_prov_attrs(process_type="processing")
    prov["processing_function"] = "clean.remove_background_noise"
    ds_Sv.attrs.update(prov)
    return insert_processing_level(ds_Sv, "L*B", input_ds=ds_Sv)


# ---- Ryan et al. (2015) noise masks ------------------------------------------------------------------

logger = logging.getLogger("echopype_amd.clean")
_CPS = ("channel", "ping_time", "range_sample")


def _cube(da, ds, dtype=None):
    """Device tensor of a variable in (channel, ping_time, range_sample) order (the reference
    transposes to it, clean/utils.py:125-127), broadcast if lower-dimensional."""
    if set(da.dims) != set(_CPS):
        da = _full(da, ds, _CPS)
    t = _dev(da, dtype)
    if tuple(da.dims) != _CPS:
        t = t.permute([da.dims.index(d) for d in _CPS])
    return t.contiguous()


def _mask_inputs(ds_Sv, range_var, need_range):
    sv_t = _cube(ds_Sv["Sv"], ds_Sv)
    if sv_t.dtype not in (torch.float32, torch.float64):
        sv_t = sv_t.double()
    rg_t = _cube(ds_Sv[range_var], ds_Sv, sv_t.dtype) if need_range else None
    return sv_t, rg_t


def _mask_da(ds_Sv, mask_t, dims=_CPS):
    """uint8 0/1 kernel output -> boolean DataArray without a copy (same item size: reinterpreted)."""
    return DataArray(DeviceArray(mask_t.view(torch.bool)), dims,
                     {d: ds_Sv[d].values for d in dims if d in ds_Sv.coords})


def _samples_per_bin(rg_t, depth_bin):
    """ceil(depth_bin / nanmean(diff(range))) per channel (clean/utils.py:131-133, :256-260)."""
    with np.errstate(invalid="ignore", divide="ignore"):
        return np.ceil(depth_bin / ops.range_step_mean(rg_t)).astype(int)


def _check_range_var(ds_Sv, range_var, what, allow_missing):
    if range_var not in ["echo_range", "depth"]:
        raise ValueError("`range_var` must be either `echo_range` or `depth`.")
    if range_var not in ds_Sv.data_vars and not allow_missing:
        raise ValueError(f"Masking {what} requires `{range_var}` data variable in `ds_Sv`.")


@xarray_io()
def mask_transient_noise(ds_Sv, func="nanmean", depth_bin="10m", num_side_pings=25, exclude_above="250.0m",
                         transient_noise_threshold="12.0dB", range_var="depth", use_index_binning=False,
                         chunk_dict={}):
    """Boolean (channel, ping_time, range_sample) mask: Sv - pooled Sv > threshold (api.py:30-168).
    ``chunk_dict`` is accepted for signature compatibility (no dask here)."""
    ds_Sv = from_xarray(ds_Sv)
    _check_range_var(ds_Sv, range_var, "transient noise", use_index_binning)
    if func != "nanmean" and func != "nanmedian":
        raise ValueError(f"Input `func` is `{func}`. `func` must be `nanmean` or `nanmedian`.")
    if func == "nanmedian":
        logger.warning(
            "`func=nanmedian` is an incredibly slow operation due to the overhead sorting. "
            "We plan to add the Fielding Transient Noise Filter in the future"
            "described here: https://github.com/OSOceanAcoustics/echopype/issues/1352")
    thr = extract_dB(transient_noise_threshold)
    depth_bin = _parse_x_bin(depth_bin, "range_bin")
    exclude_above = _parse_x_bin(exclude_above, "range_bin")
    sv_t, rg_t = _mask_inputs(ds_Sv, range_var, True)
    C, P, S = sv_t.shape
    mask = torch.empty((C, P, S), dtype=torch.uint8, device=sv_t.device)
    if not use_index_binning:
        nvalid, bad = ops.range_rows_check(rg_t)
        if bad:
            raise ValueError(f"`{range_var}` must be non-decreasing along `range_sample` with NaN only as "
                             f"trailing padding ({bad} pings are not).")
        lo, hi, _ = _range_stats(ds_Sv[range_var], rg_t)
        _, mask = ops.pool_sv_value(sv_t, rg_t, nvalid, depth_bin, num_side_pings, exclude_above, lo, hi,
                                    func=func, threshold=thr, want_pooled=False)
    else:
        n_c = _samples_per_bin(rg_t, depth_bin)
        s0 = ops.first_not_le(rg_t, exclude_above)  # np.argmin(range <= exclude_above), utils.py:143
        if s0 == rg_t.numel():
            s0 = 0
        for c in range(C):
            ops.pool_sv(sv_t[c:c + 1], s0, num_side_pings, int(n_c[c]), func=func, threshold=thr,
                        want_pooled=False, mask_out=mask[c:c + 1])  # written in place, no copy
    return _mask_da(ds_Sv, mask)


@xarray_io()
def mask_impulse_noise(ds_Sv, depth_bin="5m", num_side_pings=2, impulse_noise_threshold="10.0dB",
                       range_var="depth", use_index_binning=False):
    """Boolean impulse-noise mask (api.py:171-266); dims (channel, range_sample, ping_time), the
    order the reference's apply_ufunc leaves them in."""
    ds_Sv = from_xarray(ds_Sv)
    _check_range_var(ds_Sv, range_var, "impulse noise", use_index_binning)
    thr = extract_dB(impulse_noise_threshold)
    depth_bin = _parse_x_bin(depth_bin, "range_bin")
    sv_t, rg_t = _mask_inputs(ds_Sv, range_var, True)
    C, P, S = sv_t.shape
    if not use_index_binning:
        lo, hi, _ = _range_stats(ds_Sv[range_var], rg_t)
        nb = len(np.arange(lo, hi + depth_bin, depth_bin)) - 1
        up = ops.range_bin_smooth(sv_t, range=rg_t, r0=lo, bin=depth_bin, nbins=nb)
    else:
        n_c = _samples_per_bin(rg_t, depth_bin)
        up = torch.empty_like(sv_t)
        for c in range(C):
            ops.range_bin_smooth(sv_t[c:c + 1], nper=int(n_c[c]), out=up[c:c + 1])  # in place, no copy
    mask = ops.impulse_mask(up, num_side_pings, thr)
    # the reference's dimension order, as a transposed VIEW of the (channel, ping_time, range_sample) buffer: a
    # materialised transpose of the mask costs three times the two kernels that made it, and apply_mask permutes
    # back (to the contiguous layout) anyway
    dims = ("channel", "range_sample", "ping_time")
    return _mask_da(ds_Sv, mask.permute(0, 2, 1), dims)


@xarray_io()
def mask_attenuated_signal(ds_Sv, upper_limit_sl="400.0m", lower_limit_sl="500.0m", num_side_pings=15,
                           attenuation_signal_threshold="8.0dB", range_var="depth"):
    """Boolean (channel, ping_time, range_sample) attenuated-signal mask (api.py:269-359)."""
    ds_Sv = from_xarray(ds_Sv)
    _check_range_var(ds_Sv, range_var, "attenuated signal", False)
    if upper_limit_sl > lower_limit_sl:  # compared as given (strings), api.py:308
        raise ValueError("Minimum range has to be shorter than maximum range")
    thr = extract_dB(attenuation_signal_threshold)
    lower = _parse_x_bin(lower_limit_sl, "range_bin")
    upper = _parse_x_bin(upper_limit_sl, "range_bin")
    sv_t, rg_t = _mask_inputs(ds_Sv, range_var, True)
    lo, hi, _ = _range_stats(ds_Sv[range_var], rg_t)
    if upper > hi or lower < lo:  # searching range outside the echosounder range, api.py:322-324
        mask = torch.zeros(sv_t.shape, dtype=torch.uint8, device=sv_t.device)
    else:
        mask = ops.attenuated_mask(sv_t, rg_t, upper, lower, num_side_pings, thr)
    return _mask_da(ds_Sv, mask)


# names used by older echopype releases (docs/source/whats-new.md:366)
estimate_noise = estimate_background_noise
remove_noise = remove_background_noise
