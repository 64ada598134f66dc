"""consolidate.add_depth -- the step between compute_Sv and compute_MVBS(range_var="depth")
(SURVEY 8f "next" row 1; reference: /root/reference/echopype/consolidate/api.py:68-243).

depth = transducer_depth + orientation * echo_range * echo_range_scaling; the (channel, ping_time,
range_sample) pass is one kernel launch (epa_affine_rows), everything else is per-ping / per-channel
host work: numeric or single-dimension DataArray ``depth_offset`` / ``tilt`` (aligned to ping_time as
the reference does, utils/align.py) and the EchoData-driven options for EK60 / EK80 (platform
vertical offsets, platform angles, beam angles; consolidate/ek_depth_utils.py).
"""
import datetime
import logging
import weakref
from numbers import Number

import numpy as np
import torch

from .. import ops
from ..commongrid.api import _dev, _full
from ..xr_lite import DataArray, DeviceArray, LazyDeviceArray, from_xarray, xarray_io
from .ek_depth_utils import (align_to_ping_time, ek_use_beam_angles, ek_use_platform_angles,
                             ek_use_platform_vertical_offsets)

logger = logging.getLogger("echopype_amd.consolidate")


def _align_to_ping_time(da, ping_time):
    return align_to_ping_time(da.values, da.coords[da.dims[0]], ping_time)


def _per_channel_ping(values, dims, C, P, what):
    """Broadcast a scalar / (ping_time,) / (channel,) / (channel, ping_time) quantity to (C, P)."""
    a = np.asarray(values, dtype=np.float64)
    if a.ndim == 0:
        return np.full((C, P), float(a))
    dims = tuple(dims)
    if dims == ("ping_time",):
        return np.broadcast_to(a[None, :], (C, P))
    if dims == ("channel",):
        return np.broadcast_to(a[:, None], (C, P))
    if dims == ("channel", "ping_time"):
        return a
    if dims == ("ping_time", "channel"):
        return a.T
    raise ValueError(f"{what} has unsupported dimensions {dims}")


@xarray_io()
def swap_dims_channel_frequency(ds):
    """``frequency_nominal`` in place of ``channel`` as the dataset's dimension and coordinate (consolidate/api.py:31-65:
    set_coords -> swap_dims -> reset_coords("channel")); the reference's own compute_MVBS test bins such a dataset
    (tests/commongrid/test_commongrid_api.py:261-276).  No array is copied: the variables keep their buffers."""
    from ..xr_lite import Dataset

    ds = from_xarray(ds)
    fn = np.asarray(ds["frequency_nominal"].values)
    if np.unique(fn).size != fn.size:  # only possible if no duplicated frequencies
        raise ValueError("Duplicated transducer nominal frequencies exist in the file. Operation is not valid.")
    out = Dataset(attrs=dict(ds.attrs))
    for k, c in ds.coords.items():
        if k != "channel":
            out.coords[k] = c
    out._set_coord("frequency_nominal", DataArray(fn, ("frequency_nominal",), attrs=dict(ds["frequency_nominal"].attrs)))
    for k, v in ds.data_vars.items():
        if k != "frequency_nominal":
            dims = tuple("frequency_nominal" if d == "channel" else d for d in v.dims)
            out.data_vars[k] = DataArray(v.data, dims, attrs=dict(v.attrs), name=k)
    out["channel"] = (("frequency_nominal",), np.asarray(ds["channel"].values), dict(ds.coords["channel"].attrs))
    return out


@xarray_io(in_place=("depth",))  # the reference assigns ds["depth"] on the CALLER's dataset (consolidate/api.py:221-241)
def add_depth(ds, echodata=None, depth_offset=None, tilt=None, downward=True,
              use_platform_vertical_offsets=False, use_platform_angles=False, use_beam_angles=False):
    """Add a ``depth`` variable to an Sv dataset (in place, like the reference) and return it."""
    ds = from_xarray(ds)
    if (not echodata) and (use_platform_vertical_offsets or use_platform_angles or use_beam_angles):
        raise ValueError("If any of `use_platform_vertical_offsets`, `use_platform_angles` "
                         "or `use_beam_angles` is `True`, then `echodata` cannot be `None`.")
    if use_platform_angles and use_beam_angles:
        raise NotImplementedError("Computing depth with both platform and beam angles is not implemented yet.")
    if depth_offset is not None and use_platform_vertical_offsets:
        logger.warning("When `depth_offset` is specified, platform vertical offset "
                       "variables will not be used.")
    if tilt is not None and (use_beam_angles or use_platform_angles):
        logger.warning("When `tilt` is specified, beam/platform angle variables will not be used.")
    sonar_model = None
    if echodata:
        sonar_model = echodata["Sonar"].attrs.get("sonar_model", getattr(echodata, "sonar_model", None))
        if sonar_model not in ["EK60", "EK80"] and (
                use_platform_vertical_offsets or use_platform_angles or use_beam_angles):
            raise NotImplementedError(f"`use_platform/beam_...` not implemented yet for `{sonar_model}`.")
    depth_offset, tilt = from_xarray(depth_offset), from_xarray(tilt)

    ping_time = ds["ping_time"].values
    P = len(ping_time)
    er = ds["echo_range"]
    order = tuple(ds["Sv"].dims) if "Sv" in ds else tuple(er.dims)
    # a lazy echo_range straight from compute_Sv on power samples: its coefficient rows + the raw samples' NaN pattern go
    # to the kernel, the array is not written for this
    lazy = er.data if isinstance(er.data, LazyDeviceArray) and tuple(er.dims) == order else None
    rows = lazy.coef_rows() if lazy is not None else None
    raw = lazy.nan_source() if rows is not None else None
    if raw is not None and raw.dtype == torch.float32 and tuple(raw.shape) == lazy.shape and raw.is_contiguous():
        er_t, C = None, lazy.shape[0]
    else:
        rows = raw = None
        er_t = _dev(_full(er, ds, order))
        if er_t.dtype not in (torch.float32, torch.float64):
            er_t = er_t.double()
        C = er_t.shape[0]

    transducer_depth, td_dims = 0.0, ()
    if isinstance(depth_offset, Number):
        transducer_depth = float(depth_offset)
    if isinstance(depth_offset, DataArray):
        if len(depth_offset.dims) != 1:
            raise ValueError("If depth_offset is passed in as an xr.DataArray, it must contain a single dimension.")
        transducer_depth, td_dims = _align_to_ping_time(depth_offset, ping_time), ("ping_time",)
    elif echodata and sonar_model in ["EK60", "EK80"] and use_platform_vertical_offsets and depth_offset is None:
        transducer_depth, td_dims = ek_use_platform_vertical_offsets(echodata["Platform"], ping_time)

    scaling, sc_dims = 1.0, ()
    beam_group_name = None
    if isinstance(tilt, Number):
        scaling = float(np.cos(np.deg2rad(tilt)))
    if isinstance(tilt, DataArray):
        if len(tilt.dims) != 1:
            raise ValueError("If tilt is passed in as an xr.DataArray, it must contain a single dimension.")
        scaling, sc_dims = np.cos(np.deg2rad(_align_to_ping_time(tilt, ping_time))), ("ping_time",)
    elif echodata and sonar_model in ["EK60", "EK80"] and tilt is None:
        if use_platform_angles:
            scaling, sc_dims = ek_use_platform_angles(echodata["Platform"], ping_time)
        elif use_beam_angles:
            b1 = echodata["Sonar/Beam_group1"]
            same = np.array_equal(np.asarray(b1["channel"].values), np.asarray(ds["channel"].values))
            beam_group_name = "Beam_group1" if same else "Beam_group2"
            scaling, sc_dims = ek_use_beam_angles(echodata[f"Sonar/{beam_group_name}"])

    mult = 1.0 if downward else -1.0
    if sc_dims == () and td_dims == ():  # two numbers: filled on the device, nothing to upload
        scale_h, offset_h = np.float64(mult * scaling), np.float64(transducer_depth)
        dev = raw.device if er_t is None else er_t.device
        scale = torch.full((C, P), float(scale_h), dtype=torch.float64, device=dev)
        offset = torch.full((C, P), float(offset_h), dtype=torch.float64, device=dev)
    else:
        scale_h = mult * _per_channel_ping(scaling, sc_dims, C, P, "echo range scaling")
        offset_h = _per_channel_ping(transducer_depth, td_dims, C, P, "transducer depth")
        scale, offset = ops.to_device(np.ascontiguousarray(scale_h)), ops.to_device(np.ascontiguousarray(offset_h))
    if er_t is None:
        # echo_range is still a function of the coefficient rows: so is depth.  The array is written when somebody reads
        # it; compute_MVBS(range_var="depth") right after -- the usual sequence -- bins on it inside the pass that writes
        # Sv (epa_sv_mvbs_fused_depth) and leaves its {nanmin, nanmax, NaN count}.
        depth_data = _lazy_depth(lazy, rows, raw, scale, offset, scale_h, offset_h)
    else:
        # {nanmin, nanmax, NaN count} of depth come out of the same pass (what compute_MVBS(range_var="depth") asks next)
        depth, stats = ops.depth_rows(scale, offset, range=er_t)
        depth_data = DeviceArray(depth, stats=stats)

    used_offsets = use_platform_vertical_offsets and not _truthy(depth_offset)
    used_platform_angles = use_platform_angles and not _truthy(tilt)
    used_beam_angles = use_beam_angles and not _truthy(tilt)
    now = datetime.datetime.now(datetime.timezone.utc)
    ds["depth"] = DataArray(depth_data, order, attrs={
        "long_name": "Depth", "standard_name": "depth", "units": "m",
        "history": f"{now}. `depth` calculated using: Sv `echo_range`"
                   + (", Echodata `Platform` Vertical Offsets" if used_offsets else "")
                   + (", Echodata `Platform` Angles" if used_platform_angles else "")
                   + (f", Echodata `{beam_group_name}` Angles" if used_beam_angles and beam_group_name else "")
                   + "."})
    return ds


def _lazy_depth(er_lazy, rows, raw, scale, offset, scale_h, offset_h):
    """depth = offset + scale * echo_range of a lazy echo_range as a LazyDeviceArray: written by ``epa_depth_rows`` from
    the coefficient rows (+ the raw samples' NaN pattern) on first read, and known to be that affine function of the
    echo_range until then (``affine_of``)."""
    shape = er_lazy.shape
    tdt = torch.float64 if er_lazy.dtype == np.dtype("float64") else torch.float32

    def make():
        me = ref()
        known = me is not None and me.stats_async() is not None  # (left by the pass that binned on this depth)
        t, st = ops.depth_rows(scale, offset, coef=rows, mask_raw=raw, shape=shape, dtype=tdt, want_stats=not known)
        if me is not None and not known:
            me.set_stats(st)  # (fulfil() stamps them with the tensor's version)
        return t

    def stats_with_depth():  # statistics asked for before any pass left them: they come with the array
        me = ref()
        if me is not None:
            me.tensor

    depth_lazy = LazyDeviceArray(shape, tdt, raw.device, make)
    ref = weakref.ref(depth_lazy)  # (no cycle through the closures: a dropped dataset frees at once)
    depth_lazy.set_affine(er_lazy, scale, offset)
    depth_lazy.set_stats(None, hook=stats_with_depth)
    depth_lazy.reach_bound = _depth_reach_bound(er_lazy.reach_bound, scale_h, offset_h)
    return depth_lazy


def _depth_reach_bound(range_bound, scale, offset):
    """An upper bound of every depth value known on the host: offset + scale * echo_range with echo_range within
    [0, range_bound] (None without a bound on the range, or when nothing is finite)."""
    if range_bound is None or not np.isfinite(range_bound):
        return None
    with np.errstate(invalid="ignore"):
        hi = np.fmax.reduce(np.atleast_1d(np.fmax(offset, offset + scale * float(range_bound))), axis=None)
        hi = hi * (1 + 1e-6) if hi > 0 else hi  # (float32 depth values round up to 6e-8 above the float64 ones)
    return float(hi) if np.isfinite(hi) else None


def _truthy(v):
    """``not depth_offset`` of the reference (api.py:231-233) without the array-truthiness error."""
    return v is not None and not (isinstance(v, Number) and v == 0)
