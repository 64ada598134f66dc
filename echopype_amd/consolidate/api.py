"""consolidate.add_depth -- the step between compute_Sv and compute_MVBS(range_var="depth")
(SURVEY 8f "next" row 1; reference: /root/reference/echopype/consolidate/api.py:68-243).

depth = transducer_depth + orientation * echo_range * cos(tilt); the (channel, ping_time,
range_sample) pass is one kernel launch (epa_affine_rows).  Supported here: numeric or
single-dimension DataArray ``depth_offset`` / ``tilt`` (aligned to ping_time as the reference does,
utils/align.py).  The EchoData-driven options (platform vertical offsets, platform / beam angles,
consolidate/ek_depth_utils.py) are per-ping geometry joins outside this round's scope.
"""
import datetime
from numbers import Number

import numpy as np
import torch

from .. import ops
from ..calibrate.env_params import _interp_time
from ..commongrid.api import _dev, _full
from ..xr_lite import DataArray, DeviceArray, from_xarray


def _align_to_ping_time(da, ping_time):
    """utils/align.py:5-61 (method='nearest' in add_depth): identical axis -> as is; one value ->
    broadcast; none -> NaN; otherwise nearest-neighbour interpolation with extrapolation."""
    vals = np.asarray(da.values, dtype=np.float64)
    tname = da.dims[0]
    t = np.asarray(da.coords[tname]).astype("datetime64[ns]")
    pt = np.asarray(ping_time).astype("datetime64[ns]")
    if t.shape == pt.shape and np.array_equal(t, pt):
        return vals
    if vals.size == 1:
        return np.full(pt.shape, vals.reshape(-1)[0])
    if vals.size == 0:
        return np.full(pt.shape, np.nan)
    ti, pi = t.astype(np.int64), pt.astype(np.int64)
    idx = np.clip(np.searchsorted(ti, pi), 1, ti.size - 1)
    left_closer = (pi - ti[idx - 1]) <= (ti[idx] - pi)
    return vals[np.where(left_closer, idx - 1, idx)]


def add_depth(ds, echodata=None, depth_offset=None, tilt=None, downward=True,
              use_platform_vertical_offsets=False, use_platform_angles=False, use_beam_angles=False):
    """Add a ``depth`` variable to an Sv dataset (in place, like the reference) and return it."""
    ds = from_xarray(ds)
    if (not echodata) and (use_platform_vertical_offsets or use_platform_angles or use_beam_angles):
        raise ValueError("If any of `use_platform_vertical_offsets`, `use_platform_angles` "
                         "or `use_beam_angles` is `True`, then `echodata` cannot be `None`.")
    if use_platform_angles and use_beam_angles:
        raise NotImplementedError("Computing depth with both platform and beam angles is not implemented yet.")
    if use_platform_vertical_offsets or use_platform_angles or use_beam_angles:
        raise NotImplementedError("EchoData-driven depth offsets / angles (consolidate/ek_depth_utils.py) are "
                                  "not part of the accelerated path yet; pass depth_offset / tilt explicitly.")
    ping_time = ds["ping_time"].values
    P = len(ping_time)
    transducer_depth = np.zeros(P)
    if isinstance(depth_offset, Number):
        transducer_depth = np.full(P, float(depth_offset))
    elif isinstance(depth_offset, DataArray):
        if len(depth_offset.dims) != 1:
            raise ValueError("If depth_offset is passed in as an xr.DataArray, it must contain a single dimension.")
        transducer_depth = _align_to_ping_time(depth_offset, ping_time)
    scaling = np.ones(P)
    if isinstance(tilt, Number):
        scaling = np.full(P, np.cos(np.deg2rad(tilt)))
    elif isinstance(tilt, DataArray):
        if len(tilt.dims) != 1:
            raise ValueError("If tilt is passed in as an xr.DataArray, it must contain a single dimension.")
        scaling = np.cos(np.deg2rad(_align_to_ping_time(tilt, ping_time)))
    mult = 1.0 if downward else -1.0
    er = ds["echo_range"]
    order = tuple(ds["Sv"].dims) if "Sv" in ds else tuple(er.dims)
    er_t = _dev(_full(er, ds, order))
    if er_t.dtype not in (torch.float32, torch.float64):
        er_t = er_t.double()
    C = er_t.shape[0]
    scale = ops.to_device(np.ascontiguousarray(np.broadcast_to(mult * scaling, (C, P)), dtype=np.float64))
    offset = ops.to_device(np.ascontiguousarray(np.broadcast_to(transducer_depth, (C, P)), dtype=np.float64))
    depth = ops.affine_rows(er_t, scale, offset)
    now = datetime.datetime.now(datetime.timezone.utc)
    ds["depth"] = DataArray(DeviceArray(depth), order, attrs={
        "long_name": "Depth", "standard_name": "depth", "units": "m",
        "history": f"{now}. `depth` calculated using: Sv `echo_range`"
                   + (", user-provided `depth_offset`" if depth_offset is not None else "")
                   + (", user-provided `tilt`" if tilt is not None else "") + "."})
    return ds
