"""compute_MVBS / compute_MVBS_index_binning with the reference's signatures
(/root/reference/echopype/commongrid/api.py:30-191, :194-266).  Validation, bin edges, coordinates
and attributes are host Python (O(P)); the (channel, ping_time, range_sample) reduction is one HIP
kernel launch through the C ABI (epa_mvbs / epa_mvbs_index).  ``method``, ``reindex`` and
``**flox_kwargs`` are accepted for signature compatibility (flox is not involved).
"""
import logging
import os

import numpy as np
import torch

from .. import _lib, ops
from ..utils.prov import echopype_prov_attrs, insert_processing_level
from ..xr_lite import (DataArray, Dataset, DeferredDataset, DeviceArray, LazyDeviceArray, defer_mvbs_enabled, from_xarray,
                       xarray_io)
from .utils import (_parse_x_bin, _set_MVBS_attrs, _setup_and_validate, coarsen_time_mean, get_distance_from_latlon,
                    ping_time_bin_parsing_and_conversion, resample_edges)

logger = logging.getLogger("echopype_amd.commongrid")
_TORCH_DT = {"float64": torch.float64, "float32": torch.float32}

_AGG_MSG = ("Aggregation may be negatively impacted since Flox will not aggregate any "
            "```Sv``` values that have corresponding NaN coordinate values. Consider handling "
            "these values before calling your intended commongrid function.")


def _dev(a, dtype=None):
    data = a.data if isinstance(a, DataArray) else a
    if isinstance(data, DeviceArray):
        t = data.tensor
        return t if dtype is None or t.dtype == dtype else t.to(dtype)
    return ops.to_device(np.asarray(data), dtype=dtype)


def _range_stats(da, t):
    """(nanmin, nanmax, NaN count) of a range variable: left with the array by the kernel that wrote it
    (compute_Sv), or one sweep of the device tensor ``t`` actually handed to the kernels (None: a lazy array binned
    through its coefficient rows; it is written only if it carries no statistics)."""
    d = da.data if isinstance(da, DataArray) else None
    if isinstance(d, DeviceArray) and (t is None or (_TORCH_DT.get(d.dtype.name) == t.dtype and d.shape == tuple(t.shape))):
        st = d.cached_stats()
        if st is not None:
            return st
    return ops.nanminmax(d.tensor if t is None else t, with_nan_count=True)


def _coef_rows(da, order, sv_t):
    """The coefficient rows a lazy echo_range (xr_lite.LazyDeviceArray, left by compute_Sv on power samples) is an
    affine function of, when the binning kernels can take them in place of the array: same dims, shape and dtype as
    Sv, array untouched since.  The kernels evaluate ``fl(fl(s * ra) * rb) + r0`` rounded to the array's dtype --
    what the array holds wherever the raw sample is not NaN; where it is NaN, so is Sv, which a NaN-skipping mean
    drops like the reference drops the NaN coordinate."""
    d = da.data if isinstance(da, DataArray) else None
    if not isinstance(d, LazyDeviceArray) or tuple(da.dims) != tuple(order):
        return None
    if d.shape != tuple(sv_t.shape) or _TORCH_DT.get(d.dtype.name) != sv_t.dtype:
        return None
    return d.coef_rows()


def _full(da, ds, order):
    """Broadcast a variable to the (dim_0, ping_time, range_sample) cube if it is lower-dimensional."""
    if tuple(da.dims) == tuple(order):
        return da
    a = np.asarray(da.values)
    shape = [ds.sizes[d] for d in order]
    idx = [slice(None) if d in da.dims else None for d in order]
    src = np.transpose(a, [da.dims.index(d) for d in order if d in da.dims])
    return DataArray(np.ascontiguousarray(np.broadcast_to(src[tuple(idx)], shape)), order)


@xarray_io()
def compute_MVBS(ds_Sv, range_var="echo_range", range_bin="20m", ping_time_bin="20s", method="map-reduce",
                 reindex=False, skipna=True, fill_value=np.nan, closed="left", range_var_max=None, *, _shard=None,
                 **flox_kwargs):
    """Mean volume backscattering strength on a (ping_time, range) grid in physical units.
    (``_shard``: set by echopype_amd.sharding.compute_MVBS when ``ds_Sv`` is one rank's ping shard.)"""
    if method != "map-reduce" and reindex is not None:
        raise ValueError(f"Passing in reindex={reindex} is only allowed when method='map_reduce'.")
    ds_Sv = from_xarray(ds_Sv)
    ds_Sv, range_bin_m = _setup_and_validate(ds_Sv, range_var, range_bin, closed)
    if not isinstance(ping_time_bin, str):
        raise TypeError("ping_time_bin must be a string")

    hint = None
    if skipna and closed == "left":
        # Sv still deferred by compute_Sv: written by THIS pass over the raw samples, next to the bins ...
        # ... or the Sv_corrected remove_background_noise deferred: its pass 2 bins as well
        if _shard is None:
            done = _mvbs_of_deferred_sv(ds_Sv, range_var, range_bin_m, ping_time_bin, fill_value, range_var_max)
            if done is None:
                done = _mvbs_of_deferred_clean(ds_Sv, range_var, range_bin_m, ping_time_bin, fill_value, range_var_max)
        else:  # a ping shard: the grid is the whole dataset's, cut bins are exchanged in HBM; the route is voted on
            done, hint = _sharded_deferred(ds_Sv, range_var, range_bin_m, ping_time_bin, fill_value, range_var_max, _shard)
        if done is not None:
            return done
    return _mvbs_plain(ds_Sv, range_var, range_bin_m, ping_time_bin, skipna, fill_value, closed, range_var_max, _shard,
                       grid_hint=hint)


def _sharded_deferred(ds_Sv, range_var, range_bin_m, ping_time_bin, fill_value, range_var_max, _shard):
    """One rank's ping shard: which route the call takes is decided TOGETHER.  Whether a rank could take a deferred route
    depends on rank-local state (its Sv already read or replaced, its pings unsorted, its raw samples written to ...), and
    the routes run different collectives -- so every rank names the route it could take (1: Sv deferred by compute_Sv,
    2: Sv_corrected deferred by remove_background_noise, 0: neither) in the call's FIRST control message, the one that
    also carries the time grid and the range cap of the whole dataset (sharding.MVBSShard.grid), and a deferred route
    runs only if all ranks named it.  Returns (the MVBS dataset or None, the time grid for the plain route: it does not
    ask for it again)."""
    from .utils import timedelta_ns

    args = (ds_Sv, range_var, range_bin_m, ping_time_bin, fill_value, range_var_max, _shard)
    route, fn, prep = 0, None, None
    for rid, f in ((1, _mvbs_of_deferred_sv), (2, _mvbs_of_deferred_clean)):
        prep = f(*args, _grid="probe")
        if prep is not None:
            route, fn = rid, f
            break
    if route:
        ns, dt, r_cap = prep
    else:  # (unsorted pings, NaT, an empty shard: the message takes them as they are)
        ns = np.asarray(ds_Sv["ping_time"].values).astype("datetime64[ns]", copy=False).view(np.int64)
        dt, r_cap = timedelta_ns(ping_time_bin), float("nan")
    reach = r_cap if (route and range_var_max is None) else float("nan")
    e0g, n_glob, first_bin, last_bin, g_cap, agreed = _shard.grid(ns, dt, "left", reach, sorted_valid=bool(route), route=route)
    hint = (e0g, n_glob, first_bin, last_bin)
    if not route or agreed != route:
        return None, hint
    grid = (e0g + first_bin * dt, last_bin - first_bin + 1, first_bin, last_bin, g_cap if range_var_max is None else r_cap)
    return fn(*args, _grid=grid), hint


def _mvbs_plain(ds_Sv, range_var, range_bin_m, ping_time_bin, skipna, fill_value, closed, range_var_max, _shard,
                allow_defer=True, grid_hint=None):
    """The binning of an existing Sv array (arguments validated by compute_MVBS).  ``allow_defer=False``: the dataset is
    assembled before the call returns (the fallback of the deferred routes' own assembly).  ``grid_hint``: the time grid
    of the whole dataset, when the call's route vote has fetched it already (_sharded_deferred)."""
    sv_da = ds_Sv["Sv"]
    order = tuple(sv_da.dims)
    dim_0 = order[0]
    sv_t = _dev(sv_da)
    if sv_t.dtype not in (torch.float32, torch.float64):
        sv_t = sv_t.double()
    # (a lazy echo_range straight from compute_Sv: binned through its coefficient rows, never written)
    rows = _coef_rows(ds_Sv[range_var], order, sv_t) if skipna else None
    rg_t = _dev(_full(ds_Sv[range_var], ds_Sv, order), sv_t.dtype) if rows is None else None
    C, P, S = sv_t.shape

    if _shard is None and rows is not None and allow_defer and defer_mvbs_enabled():
        # a lazy range (coefficient rows) that knows on the host how far it can reach and whose exact statistics are a
        # kernel's by-product still in HBM: the bins are launched on the conservative grid, the dataset trimmed when read
        done = _mvbs_of_array_without_waiting(ds_Sv, sv_t, rows, range_var, range_bin_m, ping_time_bin, skipna,
                                              fill_value, closed, range_var_max)
        if done is not None:
            return done

    # range edges: np.arange(0, max + bin, bin)  (api.py:108-115)
    lo, hi, n_nan_range = _range_stats(ds_Sv[range_var], rg_t)
    if range_var_max is None:
        rmax = hi if _shard is None else _shard.range_max(hi)  # the range grid of the whole dataset
    else:
        rmax = _parse_x_bin(range_var_max) + 1e-8
    if not np.isfinite(rmax):
        raise ValueError("range bins are empty: the range variable holds no valid values")
    r_edges = np.arange(0, rmax + range_bin_m, range_bin_m)
    n_r = len(r_edges) - 1  # 0 when the only valid range is 0 (one sample per ping): an empty grid, as in the reference

    # NaN coordinates are not aggregated (utils.py:595-608: same warning text)
    ping_time = np.asarray(ds_Sv["ping_time"].values).astype("datetime64[ns]")
    if np.isnat(ping_time).any():
        logging.warning(f"The ```ping_time``` coordinate array contain NaNs. {_AGG_MSG}")
    if n_nan_range > 0:
        logging.warning(f"The ```{range_var}``` coordinate array contain NaNs. {_AGG_MSG}")

    # ping bins: pandas-resample edges, anchored at midnight (api.py:118-128)
    e0, dt, n_t = resample_edges(ping_time, ping_time_bin)
    ns = ping_time.astype(np.int64)
    first_bin = 0
    if _shard is not None:  # day origin of the whole dataset; this shard covers global bins first_bin .. last_bin
        e0, _, first_bin, last_bin = grid_hint if grid_hint is not None else _shard.time_grid(ns, dt, closed)
        e0, n_t = e0 + first_bin * dt, last_bin - first_bin + 1
    perm = None
    # unsorted pings: sort once on the host, kernels follow the permutation.  NaT is INT64_MIN: the stable sort puts
    # such pings first, below the first edge, i.e. in no bin -- flox drops values with a NaT coordinate.  (np.diff
    # would wrap in int64 on a trailing NaT and report the array as sorted.)
    if np.any(ns[1:] < ns[:-1]) or np.isnat(ping_time).any():
        order_idx = np.argsort(ns, kind="stable")
        perm = ops.to_device(order_idx.astype(np.int32))
        ns = ns[order_idx]
    bin_start = ops.time_bin_offsets(ops.to_device(ns), e0, dt, n_t, closed=closed)

    if n_r == 0:
        mvbs_t = torch.empty((C, n_t, 0), dtype=sv_t.dtype, device=sv_t.device)
    else:
        res = ops.mvbs(sv_t, bin_start, n_t, range_bin_m, n_r, range=rg_t, coef=rows, coef_as_stored=True, skipna=skipna, closed=closed,
                       fill_value=fill_value, ping_perm=perm, want_partials=_shard is not None)
        mvbs_t = res["MVBS"]
        if _shard is not None:  # bins cut by a shard edge: totals over all ranks, reported by the lowest holder
            mvbs_t, lo = _shard.finish(res, first_bin, last_bin, fill_value)
            e0, n_t = e0 + lo * dt, mvbs_t.shape[1]

    return _assemble_mvbs(ds_Sv, mvbs_t, dim_0, ping_time, e0, dt, n_t, r_edges, range_var, range_bin_m,
                          ping_time_bin, closed)


def _mvbs_of_array_without_waiting(ds_Sv, sv_t, rows, range_var, range_bin_m, ping_time_bin, skipna, fill_value, closed,
                                   range_var_max):
    """The binning of an Sv ARRAY whose range variable is still lazy (``compute_Sv`` on EK80 complex samples; an EK60 Sv
    somebody has read already) -- without the wait ``_mvbs_plain`` pays for nanmax(range) before it can size its grid:
    the kernel runs on ``np.arange(0, bound + bin, bin)`` with the host-side bound the lazy range carries
    (``reach_bound``) and the dataset, a ``DeferredDataset``, is trimmed to ``nanmax`` when somebody reads it.  None: the
    plain route (no bound, no statistics, unsorted pings, a grid the kernels refuse)."""
    rng_d = ds_Sv[range_var].data
    if not isinstance(rng_d, LazyDeviceArray):
        return None
    bound = (_parse_x_bin(range_var_max) + 1e-8) if range_var_max is not None else rng_d.reach_bound
    if bound is None or not np.isfinite(bound):
        return None
    ping_time = np.asarray(ds_Sv["ping_time"].values).astype("datetime64[ns]", copy=False)
    ns, sorted_valid, _ = ops.ping_time_facts(ping_time, want_device=False)
    if not sorted_valid:  # unsorted pings, NaT
        return None
    n_cap = len(np.arange(0, bound + range_bin_m, range_bin_m)) - 1
    stats = rng_d.stats_async() if n_cap >= 1 else None
    if stats is None:
        return None
    C, P, S = sv_t.shape
    e0, dt, n_t = resample_edges(ping_time, ping_time_bin, sorted_valid=True)
    bin_start = ops.time_bin_offsets(ops.ping_time_facts(ping_time)[2], e0, dt, n_t, closed=closed)
    try:
        res = ops.mvbs(sv_t, bin_start, n_t, range_bin_m, n_cap, coef=rows, coef_as_stored=True, skipna=skipna,
                       closed=closed, fill_value=fill_value)
    except _lib.EpaError:
        return None
    dim_0 = tuple(ds_Sv["Sv"].dims)[0]
    # what the assembly needs is taken NOW: the caller may replace variables or edit attributes of its dataset between
    # this call and the first use of the result, and the result must not change with them; the kernel's partial sums /
    # counts are not kept alive by the closure
    ds_Sv, mv_full = ds_Sv.copy(), res["MVBS"]
    del res

    def build():
        lo, hi, n_nan_range = stats.tolist()
        rmax = hi if range_var_max is None else bound
        r_edges = np.arange(0, rmax + range_bin_m, range_bin_m) if np.isfinite(rmax) else np.zeros(1)
        n_r = len(r_edges) - 1
        if n_r < 1 or n_r > n_cap:  # (no valid range / an empty grid: the plain route raises or returns the reference's)
            return _mvbs_plain(ds_Sv, range_var, range_bin_m, ping_time_bin, skipna, fill_value, closed, range_var_max, None, allow_defer=False)
        if n_nan_range > 0:
            logging.warning(f"The ```{range_var}``` coordinate array contain NaNs. {_AGG_MSG}")
        mvbs_t = mv_full[..., :n_r].contiguous() if n_r != n_cap else mv_full
        return _assemble_mvbs(ds_Sv, mvbs_t, dim_0, ping_time, e0, dt, n_t, r_edges, range_var, range_bin_m,
                              ping_time_bin, closed)

    return DeferredDataset(build)


def _shard_assemble(ds_Sv, mv_full, n_cap, rmax, r_cap, range_var_max, n_nan_range, ping_time, e0, dt, n_t, range_var,
                    range_bin_m, ping_time_bin):
    """The deferred assembly on a ping shard: ``rmax`` = nanmax(range) over ALL shards (all-reduced in HBM).  No
    fallback here -- a fallback would be a collective at a time only this rank chooses."""
    rmax = rmax if range_var_max is None else r_cap
    if not np.isfinite(rmax):  # no valid range on any rank: the reference's grid does not exist
        raise ValueError("range bins are empty: the range variable holds no valid values")
    r_edges = np.arange(0, rmax + range_bin_m, range_bin_m)
    n_r = len(r_edges) - 1
    if n_r > n_cap:
        raise RuntimeError(f"nanmax({range_var}) = {rmax} lies beyond the range grid the shards agreed on "
                           f"({n_cap} bins of {range_bin_m} m)")
    if n_nan_range > 0:
        logging.warning(f"The ```{range_var}``` coordinate array contain NaNs. {_AGG_MSG}")
    mvbs_t = mv_full[..., :n_r].contiguous() if n_r != n_cap else mv_full
    return _assemble_mvbs(ds_Sv, mvbs_t, "channel", ping_time, e0, dt, n_t, r_edges, range_var, range_bin_m,
                          ping_time_bin, "left")


def _mvbs_of_deferred_sv(ds_Sv, range_var, range_bin_m, ping_time_bin, fill_value, range_var_max, _shard=None, _grid=None):
    """``compute_Sv`` on power samples leaves Sv deferred (``LazyDeviceArray`` with a ``source``); binned right after --
    the usual sequence -- one pass over the raw samples (``epa_sv_mvbs_fused``) writes that Sv array AND the bins: the two
    reference calls cost 12 B/sample instead of 12 + 8.  Returns the MVBS dataset, or None when the plain route has to
    run (Sv already written or replaced, another range variable, unsorted / NaT pings, a grid the fused kernel does not
    serve); when the pass ran, its Sv array and range statistics stay with the dataset either way.

    Nothing here waits for the GPU: the kernel runs on a conservative range grid (the farthest any coefficient row can
    reach, bounded on the host from the host copies of sample_interval / sound_speed when there are any), leaves
    {nanmin, nanmax, NaN count} of the echo_range in HBM, and the dataset that needs them -- the grid is
    ``np.arange(0, nanmax + bin, bin)`` (api.py:108-115) -- is a ``DeferredDataset``: read back, trimmed and assembled
    on first use."""
    sv_da, rng_da = ds_Sv["Sv"], ds_Sv[range_var]
    d = sv_da.data
    src = d.source if isinstance(d, LazyDeviceArray) and not d.materialized else None
    dims = ("channel", "ping_time", "range_sample")
    # ``depth`` left lazy by add_depth on this dataset's lazy echo_range (offset + scale * echo_range, row by row): binned
    # inside the same pass (epa_sv_mvbs_fused_depth) -- the three reference calls at the 12 B/sample of the two
    rng_d, depth = rng_da.data, None
    if src is not None and rng_d is not getattr(src, "echo_range", None) and isinstance(rng_d, LazyDeviceArray):
        aff = rng_d.affine_of()
        if aff is not None and aff[0] is getattr(src, "echo_range", None) \
                and src.flags == (_lib.FLAG_GUARD_POS | _lib.FLAG_MASK_RANGE) and rng_d.dtype == d.dtype:
            depth = aff[1:]
    if src is None or getattr(src, "cal_type", None) != "Sv" or (depth is None and rng_d is not src.echo_range) \
            or tuple(sv_da.dims) != dims \
            or tuple(rng_da.dims) != dims or src.echo_range.coef_rows() is not src.coef or not src.intact():
        return None  # (also: an Sv deferred by something else than compute_Sv, e.g. remove_background_noise)
    ping_time = np.asarray(ds_Sv["ping_time"].values).astype("datetime64[ns]", copy=False)
    # (kept with the array when it is resident data, EchoData.to_device: no O(P) pass, no upload per call)
    ns, sorted_valid, _ = ops.ping_time_facts(ping_time, want_device=False)
    if not sorted_valid:  # unsorted, NaT, or no ping at all
        return None
    e0, dt, n_t = resample_edges(ping_time, ping_time_bin, sorted_valid=True)
    C, P, S = d.shape
    if range_var_max is not None:
        r_cap = _parse_x_bin(range_var_max) + 1e-8
    elif depth is not None and rng_d.reach_bound is not None:
        r_cap = rng_d.reach_bound
    elif depth is None and getattr(src, "reach_bound", None) is not None:
        r_cap = src.reach_bound  # host-side bound: no device reduction, no wait
    else:
        coef = src.coef
        reach = (S - 1) * coef[..., _lib.CF_RA] * coef[..., _lib.CF_RB] + coef[..., _lib.CF_R0]
        if depth is not None:  # offset + scale * [r0, reach], whichever end is the deeper one
            reach = torch.fmax(depth[1] + depth[0] * reach, depth[1] + depth[0] * coef[..., _lib.CF_R0]) * (1 + 1e-6)
        r_cap = float(torch.nan_to_num(reach, nan=float("-inf")).max().item())
    first_bin = last_bin = 0
    if _shard is not None:
        if isinstance(_grid, str):  # "probe": this rank could take the route (_sharded_deferred puts it to the vote)
            return ns, dt, r_cap
        e0, n_t, first_bin, last_bin, r_cap = _grid  # (the whole dataset's: the same on every rank)
    n_cap = len(np.arange(0, r_cap + range_bin_m, range_bin_m)) - 1 if np.isfinite(r_cap) else 0
    if n_cap < 1:
        return None
    bin_start = ops.time_bin_offsets(ops.ping_time_facts(ping_time)[2], e0, dt, n_t)
    res = None
    try:
        if depth is not None:
            # EPA_DEPTH_WITH_MVBS=1: the depth array is written by this pass as well (+8 B/sample here instead of the
            # 4 + 8 of epa_depth_rows when somebody reads it later); default: it stays lazy
            res = ops.sv_mvbs_fused_depth(src.raw, src.coef, depth[0], depth[1], bin_start, n_t, range_bin_m, n_cap,
                                          fill_value=fill_value, dtype=src.dtype, want_partials=_shard is not None,
                                          want_depth=not rng_d.materialized and os.environ.get("EPA_DEPTH_WITH_MVBS") == "1")
        else:
            res = ops.sv_mvbs_fused(src.raw, src.coef, bin_start, n_t, range_bin_m, n_cap, cal_flags=src.flags, skipna=True,
                                    closed="left", fill_value=fill_value, dtype=src.dtype, want_range_stats=True,
                                    want_partials=_shard is not None)
    except _lib.EpaError:  # e.g. a range grid too fine for the LDS accumulators
        pass
    # served by the generic kernel (a handful of pings, a grid beyond the LDS)?  It leaves no range statistics, which
    # every later step asks for -- K1 does, on the plain route.  (Known on the host: epa_last_range_stats_filled.)
    declined = res is None or not res["range_stats_filled"]
    done = None
    if _shard is not None:  # the plain route runs other collectives: every rank takes it if any has to -- the vote rides
        done = _shard.finish(None if declined else res, first_bin, last_bin, fill_value,  # with the exchange plan's message
                             shape=(C, n_t, n_cap), device=src.raw.device)
        declined = done is None
    if declined:
        return None
    rng = rng_d if depth is not None else src.echo_range  # (the statistics are the binned variable's)
    d.fulfil(res["Sv"])
    if res.get("depth") is not None:
        rng_d.fulfil(res["depth"])
    # the three numbers start their way to the host now, on a side stream behind THIS kernel: whoever reads them later
    # (the assembly below, the next reader of the echo_range statistics) does not wait for kernels launched after it
    stats = ops.fetch_async(res["range_stats"])
    rng.set_stats(stats)
    if _shard is not None:
        # nanmax(echo_range) of the whole dataset: all-reduced (MAX) where it lies, behind the kernel; cut bins: totals
        # over all ranks, reported by the lowest holder (one all-reduce in HBM) -- the host waits for neither
        gmax = ops.fetch_async(_shard.range_max_device(res["range_stats"][1:2].clone())) if range_var_max is None else None
        mv_full, lo = done
        e0, n_t = e0 + lo * dt, mv_full.shape[1]
        ds_Sv = ds_Sv.copy()
        del res, done

        def build_shard():
            return _shard_assemble(ds_Sv, mv_full, n_cap, gmax.item() if gmax is not None else r_cap, r_cap,
                                   range_var_max, stats.tolist()[2], ping_time, e0, dt, n_t, range_var, range_bin_m,
                                   ping_time_bin)

        return DeferredDataset(build_shard) if defer_mvbs_enabled() else build_shard()
    ds_Sv, mv_full = ds_Sv.copy(), res["MVBS"]  # (snapshot: see _mvbs_of_array_without_waiting)
    del res

    def build():
        lo, hi, n_nan_range = stats.tolist()
        rmax = hi if range_var_max is None else r_cap
        r_edges = np.arange(0, rmax + range_bin_m, range_bin_m) if np.isfinite(rmax) else np.zeros(1)
        n_r = len(r_edges) - 1
        # no valid range / an empty grid: the plain route raises or returns what the reference would; a maximum beyond the
        # conservative grid the kernel ran on (it must not happen: the bound is an upper one): binned again, exactly
        if n_r < 1 or n_r > n_cap:
            return _mvbs_plain(ds_Sv, range_var, range_bin_m, ping_time_bin, True, fill_value, "left", range_var_max, None, allow_defer=False)
        if n_nan_range > 0:
            logging.warning(f"The ```{range_var}``` coordinate array contain NaNs. {_AGG_MSG}")
        mvbs_t = mv_full[..., :n_r].contiguous() if n_r != n_cap else mv_full
        return _assemble_mvbs(ds_Sv, mvbs_t, "channel", ping_time, e0, dt, n_t, r_edges, range_var, range_bin_m,
                              ping_time_bin, "left")

    return DeferredDataset(build) if defer_mvbs_enabled() else build()


def _mvbs_of_deferred_clean(ds_Sv, range_var, range_bin_m, ping_time_bin, fill_value, range_var_max, _shard=None, _grid=None):
    """The chain as the reference's user writes it -- ``compute_Sv``, ``remove_background_noise``, then
    ``compute_MVBS`` of the dataset with ``Sv := Sv_corrected`` -- arrives here with an Sv that
    ``remove_background_noise`` left deferred (``clean.api.DenoiseSource``: pass 1 has run).  Pass 2 runs NOW, on the
    real grid: one sweep of the raw samples writes Sv_noise, Sv_corrected, their minima / maxima AND the bins
    (``epa_sv_denoise_mvbs``) -- 20 B/sample instead of 20 + 8 for the pass and a separate binning of the array it
    wrote.  Nothing waits for the GPU (conservative grid, ``DeferredDataset``: see ``_mvbs_of_deferred_sv``).  None:
    the plain route runs (it reads the array, which runs pass 2 by itself)."""
    from ..clean.api import DenoiseSource

    sv_da, rng_da = ds_Sv["Sv"], ds_Sv[range_var]
    d = sv_da.data
    src = d.source if isinstance(d, LazyDeviceArray) and not d.materialized else None
    dims = ("channel", "ping_time", "range_sample")
    if not isinstance(src, DenoiseSource) or d is not src.lazy("corrected") or src.minmax is not None or not src.intact():
        return None
    p = src.power
    if rng_da.data is not p.echo_range or tuple(sv_da.dims) != dims or tuple(rng_da.dims) != dims \
            or p.echo_range.coef_rows() is not p.coef:
        return None
    ping_time = np.asarray(ds_Sv["ping_time"].values).astype("datetime64[ns]", copy=False)
    ns, sorted_valid, _ = ops.ping_time_facts(ping_time, want_device=False)
    if not sorted_valid:  # unsorted pings, NaT
        return None
    e0, dt, n_t = resample_edges(ping_time, ping_time_bin, sorted_valid=True)
    C, P, S = d.shape
    if range_var_max is not None:
        r_cap = _parse_x_bin(range_var_max) + 1e-8
    elif getattr(p, "reach_bound", None) is not None:
        r_cap = p.reach_bound
    else:
        coef = p.coef
        reach = torch.nan_to_num((S - 1) * coef[..., _lib.CF_RA] * coef[..., _lib.CF_RB] + coef[..., _lib.CF_R0], nan=float("-inf"))
        r_cap = float(reach.max().item())
    first_bin = last_bin = 0
    if _shard is not None:
        if src.global_rmax is None:  # (pass 1 did not run as a shard's: the plain route -- by the vote, if on this rank only)
            return None
        if isinstance(_grid, str):  # "probe": see _mvbs_of_deferred_sv
            return ns, dt, r_cap
        e0, n_t, first_bin, last_bin, r_cap = _grid
    n_cap = len(np.arange(0, r_cap + range_bin_m, range_bin_m)) - 1 if np.isfinite(r_cap) else 0
    if n_cap < 1:
        return None
    bin_start = ops.time_bin_offsets(ops.ping_time_facts(ping_time)[2], e0, dt, n_t)
    res = None
    try:
        res = ops.sv_denoise_mvbs(p.raw, p.coef, src.a2, src.noise, src.ping_num, float(src.snr), bin_start, n_t,
                                  range_bin_m, n_cap, flags=p.flags, dtype=p.dtype, skipna=True, closed="left",
                                  fill_value=fill_value, want_noise=True, want_corrected=True, want_minmax=True,
                                  minmax_async=True, ping_phase=src.ping_phase, want_partials=_shard is not None)
    except _lib.EpaError:  # e.g. a range grid too fine for the LDS accumulators
        pass
    done = None
    if _shard is not None:  # (the vote on the fallback rides with the exchange plan's message)
        done = _shard.finish(res, first_bin, last_bin, fill_value, shape=(C, n_t, n_cap), device=p.raw.device)
    if (done is None) if _shard is not None else res is None:
        return None
    src.install(res)
    rng = p.echo_range
    if _shard is not None:  # cut bins: totals over all ranks, reported by the lowest holder (one all-reduce in HBM)
        mv_full, lo = done
        e0, n_t = e0 + lo * dt, mv_full.shape[1]
        ds_Sv, gmax = ds_Sv.copy(), src.global_rmax
        del res, done

        def build_shard():
            st = rng.cached_stats()  # (local: the NaN-coordinate warning)
            rmax = float(gmax.item())
            return _shard_assemble(ds_Sv, mv_full, n_cap, rmax if rmax > float("-inf") else float("nan"), r_cap,
                                   range_var_max, st[2] if st is not None else 0, ping_time, e0, dt, n_t, range_var,
                                   range_bin_m, ping_time_bin)

        return DeferredDataset(build_shard) if defer_mvbs_enabled() else build_shard()
    ds_Sv, mv_full = ds_Sv.copy(), res["MVBS"]  # (snapshot: see _mvbs_of_array_without_waiting)
    del res

    def build():
        stats = rng.cached_stats()  # left by pass 1, on their way to the host since then
        if stats is None:
            return _mvbs_plain(ds_Sv, range_var, range_bin_m, ping_time_bin, True, fill_value, "left", range_var_max, None, allow_defer=False)
        lo, hi, n_nan_range = stats
        rmax = hi if range_var_max is None else r_cap
        r_edges = np.arange(0, rmax + range_bin_m, range_bin_m) if np.isfinite(rmax) else np.zeros(1)
        n_r = len(r_edges) - 1
        if n_r < 1 or n_r > n_cap:
            return _mvbs_plain(ds_Sv, range_var, range_bin_m, ping_time_bin, True, fill_value, "left", range_var_max, None, allow_defer=False)
        if n_nan_range > 0:
            logging.warning(f"The ```{range_var}``` coordinate array contain NaNs. {_AGG_MSG}")
        mvbs_t = mv_full[..., :n_r].contiguous() if n_r != n_cap else mv_full
        return _assemble_mvbs(ds_Sv, mvbs_t, "channel", ping_time, e0, dt, n_t, r_edges, range_var, range_bin_m,
                              ping_time_bin, "left")

    return DeferredDataset(build) if defer_mvbs_enabled() else build()


def _assemble_mvbs(ds_Sv, mvbs_t, dim_0, ping_time, e0, dt, n_t, r_edges, range_var, range_bin_m,
                   ping_time_bin, closed):
    """Coordinates (left bin edges), positions, attributes and provenance of the MVBS dataset
    (commongrid/api.py:146-189)."""
    t_left = (e0 + dt * np.arange(n_t)).astype("datetime64[ns]")
    ds_MVBS = Dataset(coords={"ping_time": t_left, dim_0: ds_Sv[dim_0].values, range_var: r_edges[:-1]})
    ds_MVBS["Sv"] = DataArray(DeviceArray(mvbs_t), (dim_0, "ping_time", range_var))

    # positions: per-bin nanmean of latitude / longitude (utils.py:453-501); O(P) on the host
    if "latitude" in ds_Sv and "longitude" in ds_Sv:
        tb = np.floor_divide(ping_time.astype(np.int64) - e0, dt)
        if closed == "right":
            tb = -np.floor_divide(-(ping_time.astype(np.int64) - e0), dt) - 1
        ok = (tb >= 0) & (tb < n_t) & ~np.isnat(ping_time)
        for var in ("latitude", "longitude"):
            v = np.asarray(ds_Sv[var].values, dtype=np.float64)
            use = ok & ~np.isnan(v)
            ssum = np.bincount(tb[use], weights=v[use], minlength=n_t)
            cnt = np.bincount(tb[use], minlength=n_t)
            with np.errstate(invalid="ignore", divide="ignore"):
                ds_MVBS[var] = (("ping_time",), np.where(cnt > 0, ssum / cnt, np.nan), dict(ds_Sv[var].attrs))
    if range_var == "echo_range" and "water_level" in ds_Sv.data_vars:
        ds_MVBS["water_level"] = ds_Sv["water_level"]

    _set_MVBS_attrs(ds_MVBS)
    ds_MVBS.coords[range_var].attrs = {"long_name": "Range distance", "units": "m"}
    val, unit = ping_time_bin_parsing_and_conversion(ping_time_bin)
    ds_MVBS.data_vars["Sv"].attrs.update({
        "cell_methods": (f"ping_time: mean (interval: {val} {unit} "
                         "comment: ping_time is the interval start) "
                         f"{range_var}: mean (interval: {range_bin_m} meter "
                         f"comment: {range_var} is the interval start)"),
        "binning_mode": "physical units",
        "range_meter_interval": str(range_bin_m) + "m",
        "ping_time_interval": ping_time_bin,
    })
    prov = echopype_prov_attrs(process_type="processing")
    prov["processing_function"] = "commongrid.compute_MVBS"
    ds_MVBS = ds_MVBS.assign_attrs(prov)
    if "frequency_nominal" in ds_Sv:
        ds_MVBS["frequency_nominal"] = ds_Sv["frequency_nominal"]
    if "channel" in ds_Sv and dim_0 != "channel":
        ds_MVBS["channel"] = ds_Sv["channel"]
    return insert_processing_level(ds_MVBS, "L3*", input_ds=ds_Sv)


@xarray_io()
def compute_MVBS_index_binning(ds_Sv, range_sample_num=100, ping_num=100):
    """MVBS over blocks of ``ping_num`` pings x ``range_sample_num`` samples (api.py:194-266)."""
    ds_Sv = from_xarray(ds_Sv)
    sv_da = ds_Sv["Sv"]
    order = tuple(sv_da.dims)
    sv_t = _dev(sv_da)
    if sv_t.dtype not in (torch.float32, torch.float64):
        sv_t = sv_t.double()
    rg_t = _dev(_full(ds_Sv["echo_range"], ds_Sv, order), sv_t.dtype)
    mv, rmin = ops.mvbs_index(sv_t, ping_num, range_sample_num, range=rg_t)
    C, Pb, Sb = mv.shape
    ping_time = np.asarray(ds_Sv["ping_time"].values)
    ds_MVBS = Dataset(coords={order[0]: ds_Sv[order[0]].values, "ping_time": coarsen_time_mean(ping_time, ping_num)[:Pb],
                              "range_sample": np.arange(Sb)})
    ds_MVBS.coords["range_sample"].attrs = {"long_name": "Along-range sample number, base 0"}
    ds_MVBS["Sv"] = DataArray(DeviceArray(mv), (order[0], "ping_time", "range_sample"))
    ds_MVBS["echo_range"] = DataArray(DeviceArray(rmin), (order[0], "ping_time", "range_sample"))
    _set_MVBS_attrs(ds_MVBS)
    lo, hi = ops.nanminmax(mv)
    ds_MVBS.data_vars["Sv"].attrs.update({
        "cell_methods": (f"ping_time: mean (interval: {ping_num} pings "
                         "comment: ping_time is the interval start) "
                         f"range_sample: mean (interval: {range_sample_num} samples along range "
                         "comment: range_sample is the interval start)"),
        "comment": "MVBS binned on the basis of range_sample and ping number specified as index numbers",
        "binning_mode": "sample number",
        "range_sample_interval": f"{range_sample_num} samples along range",
        "ping_interval": f"{ping_num} pings",
        "actual_range": [round(float(lo), 2), round(float(hi), 2)],
    })
    prov = echopype_prov_attrs(process_type="processing")
    prov["processing_function"] = "commongrid.compute_MVBS_index_binning"
    ds_MVBS = ds_MVBS.assign_attrs(prov)
    if "frequency_nominal" in ds_Sv:
        ds_MVBS["frequency_nominal"] = ds_Sv["frequency_nominal"]
    return insert_processing_level(ds_MVBS, "L3*", input_ds=ds_Sv)


POSITION_VARIABLES = ["latitude", "longitude"]


@xarray_io()
def compute_NASC(ds_Sv, range_bin="10m", dist_bin="0.5nmi", method="map-reduce", skipna=True, closed="left",
                 **flox_kwargs):
    """Nautical Areal Scattering Coefficient on a (distance, depth) grid (api.py:269-416).
    ``ds_Sv`` must hold ``Sv``, ``depth``, ``latitude`` and ``longitude``; it may be an Sv dataset
    (dims channel, ping_time, range_sample) or an MVBS dataset gridded on depth."""
    ds_Sv = from_xarray(ds_Sv)
    range_var = "depth"
    ds_Sv, range_bin_m = _setup_and_validate(ds_Sv, range_var, range_bin, closed,
                                             required_data_vars=POSITION_VARIABLES)
    if not isinstance(dist_bin, str):
        raise TypeError("dist_bin must be a string")
    dist_bin_nmi = _parse_x_bin(dist_bin, "dist_bin")

    dist_nmi = get_distance_from_latlon(ds_Sv)  # O(P) on the host
    sv_da = ds_Sv["Sv"]
    order = tuple(sv_da.dims)
    dim_0 = order[0]
    sv_t = _dev(sv_da)
    if sv_t.dtype not in (torch.float32, torch.float64):
        sv_t = sv_t.double()
    dp_t = _dev(_full(ds_Sv[range_var], ds_Sv, order), sv_t.dtype)
    C, P, S = sv_t.shape

    lo, hi, _ = _range_stats(ds_Sv[range_var], dp_t)
    r_edges = np.arange(0, hi + range_bin_m, range_bin_m)
    d_edges = np.arange(0, np.nanmax(dist_nmi) + dist_bin_nmi, dist_bin_nmi)
    n_r, n_d = len(r_edges) - 1, len(d_edges) - 1
    if n_r < 1 or n_d < 1:
        raise ValueError("NASC bins are empty: depth / distance hold no valid values")
    # cumulative distance is non-decreasing: the pings of a distance bin are contiguous
    side = "left" if closed == "left" else "right"
    starts = np.searchsorted(dist_nmi, d_edges, side=side).astype(np.int32)
    bin_start = ops.to_device(starts)

    nasc_t = ops.nasc(sv_t, dp_t, bin_start, n_d, range_bin_m, n_r, skipna=skipna, closed=closed)

    ds_NASC = Dataset(coords={"distance": d_edges[:-1], dim_0: ds_Sv[dim_0].values, range_var: r_edges[:-1]})
    ds_NASC["NASC"] = DataArray(DeviceArray(nasc_t), (dim_0, "distance", range_var))
    # per distance bin: mean position and mean ping_time (utils.py:453-501, :162-171)
    idx = np.searchsorted(d_edges, dist_nmi, side="right" if closed == "left" else "left") - 1
    inb = (idx >= 0) & (idx < n_d)
    for var in POSITION_VARIABLES:
        v = np.asarray(ds_Sv[var].values, dtype=np.float64)
        use = inb & ~np.isnan(v)
        ssum = np.bincount(idx[use], weights=v[use], minlength=n_d)
        cnt = np.bincount(idx[use], minlength=n_d)
        with np.errstate(invalid="ignore", divide="ignore"):
            ds_NASC[var] = (("distance",), np.where(cnt > 0, ssum / cnt, np.nan), dict(ds_Sv[var].attrs))
    ping_time = np.asarray(ds_Sv["ping_time"].values).astype("datetime64[ns]")
    ns = ping_time.astype(np.int64)
    valid_t = inb & ~np.isnat(ping_time)
    n_t = np.bincount(idx[valid_t], minlength=n_d)
    t0 = ns[valid_t].min() if valid_t.any() else 0
    tsum = np.bincount(idx[valid_t], weights=(ns[valid_t] - t0).astype(np.float64), minlength=n_d)
    with np.errstate(invalid="ignore", divide="ignore"):
        tmean = np.where(n_t > 0, t0 + np.round(tsum / np.maximum(n_t, 1)), np.iinfo(np.int64).min)
    ds_NASC["ping_time"] = (("distance",), tmean.astype(np.int64).astype("datetime64[ns]"),
                            dict(ds_Sv["ping_time"].attrs))
    if "frequency_nominal" in ds_Sv:
        ds_NASC["frequency_nominal"] = ds_Sv["frequency_nominal"]

    ds_NASC.data_vars["NASC"].attrs = {"long_name": "Nautical Areal Scattering Coefficient (NASC, m2 nmi-2)",
                                       "units": "m2 nmi-2"}
    ds_NASC.coords["distance"].attrs = {"long_name": "Cumulative distance", "units": "nmi"}
    ds_NASC.coords["depth"].attrs = {"long_name": "Cell depth", "units": "m", "standard_name": "depth"}
    tv = ping_time[~np.isnat(ping_time)]
    lat = np.asarray(ds_Sv["latitude"].values, dtype=np.float64)
    lon = np.asarray(ds_Sv["longitude"].values, dtype=np.float64)
    ds_NASC.attrs.update({
        "Conventions": "CF-1.7,ACDD-1.3",
        "time_coverage_start": np.datetime_as_string(tv.min(), timezone="UTC"),
        "time_coverage_end": np.datetime_as_string(tv.max(), timezone="UTC"),
        "geospatial_lat_min": round(float(np.nanmin(lat)), 5),
        "geospatial_lat_max": round(float(np.nanmax(lat)), 5),
        "geospatial_lon_min": round(float(np.nanmin(lon)), 5),
        "geospatial_lon_max": round(float(np.nanmax(lon)), 5),
    })
    return insert_processing_level(ds_NASC, "L4", input_ds=ds_Sv)
