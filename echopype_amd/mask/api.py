"""apply_mask with the reference's signature (/root/reference/echopype/mask/api.py:307-464;
SURVEY 8f row 2).  Validation is host Python with the reference's error types and messages
(:39-70 dim alignment, :72-190 mask input, :193-247 var_name / fill_value); the array work is
epa_mask_and (logical AND of broadcast masks, :402-408) and epa_apply_mask (xr.where, :428-432).
Masks and sources given as file paths are out of scope (no zarr / netCDF IO in this package).
"""
import datetime
import pathlib

import numpy as np
import torch

from .. import ops
from ..commongrid.api import _dev
from ..utils.prov import echopype_prov_attrs, insert_processing_level
from ..xr_lite import DataArray, DeviceArray, from_xarray, is_device, xarray_io

_ALLOWED_DIMS = [
    {"ping_time", "range_sample"}, {"ping_time", "depth"}, {"ping_time", "echo_range"},
    {"channel", "ping_time", "range_sample"}, {"channel", "ping_time", "depth"},
    {"channel", "ping_time", "echo_range"},
]


def _no_paths(obj, what):
    if isinstance(obj, (str, pathlib.Path)):
        raise NotImplementedError(f"{what} given as a file path is not supported: pass the Dataset / DataArray")


def _validate_and_collect_mask_input(mask, storage_options_mask):
    single = not isinstance(mask, list)
    if single:
        mask = [mask]
        if not isinstance(storage_options_mask, dict):
            raise ValueError("The provided input storage_options_mask should be a single "
                             "dict because mask is a single value!")
        storage_options_mask = [storage_options_mask]
    if not isinstance(storage_options_mask, list):
        if not isinstance(storage_options_mask, dict):
            raise TypeError("storage_options_mask must be a list of dict or a dict!")
    elif not all(isinstance(e, dict) for e in storage_options_mask):
        raise TypeError("storage_options_mask must be a list of dict or a dict!")
    out = []
    for m in mask:
        _no_paths(m, "mask")
        m = from_xarray(m)
        if not isinstance(m, DataArray):
            raise TypeError("mask must be a DataArray or a list of DataArrays")
        if set(m.dims) not in _ALLOWED_DIMS:
            raise ValueError(
                "Masks must have one of the following dimensions: "
                "{'ping_time', 'range_sample'}, {'ping_time', 'depth'}, {'ping_time', 'echo_range'}, "
                "{'channel', 'ping_time', 'range_sample'}, {'channel', 'ping_time', 'depth'}"
                "{'channel', 'ping_time', 'echo_range'}")
        # boolean-like values only (:166-175); bool / uint8 device masks from this package's own
        # mask functions are boolean by construction
        if not (is_device(m.data) and m.data.tensor.dtype == torch.bool) and m.dtype != np.bool_:
            v = m.values
            if np.issubdtype(v.dtype, np.floating) and np.any(np.isnan(v)):
                raise TypeError("Mask cannot contain NaN")
            if not np.all(np.isin(np.unique(v), [0, 1, True, False])):
                raise TypeError("Mask must be boolean (True/False or 1/0)")
        out.append(m)
    shapes = set()
    for m in out:
        if "channel" in m.dims:
            # by dimension NAME: the reference compares positional shapes and so rejects its own
            # (channel, range_sample, ping_time) impulse mask next to a (channel, ping_time, ...) one
            shapes.add(frozenset((d, n) for d, n in zip(m.dims, m.shape) if d != "channel"))
    if len(shapes) > 1:
        raise ValueError("All masks must have the same shape in the 'channel' dimension.")
    return out[0] if single else out


def _check_mask_dim_alignment(source_ds, mask, var_name):
    masks = mask if isinstance(mask, list) else [mask]
    mask_dims = set()
    for m in masks:
        mask_dims.update(m.dims)
    target = set(source_ds[var_name].dims)
    if "channel" in mask_dims and "channel" not in target:
        raise ValueError("'channel' is a dimension in mask but not a dimension in source.")
    mask_dims.discard("channel")
    target.discard("channel")
    if mask_dims != target:
        raise ValueError(f"The dimensions of mask: ({mask_dims}) do not match the dimensions of source "
                         f"({target}) when not considering 'channel'.")
    return source_ds


def _check_var_name_fill_value(source_ds, var_name, fill_value):
    if not isinstance(var_name, str):
        raise TypeError("The input var_name must be a string!")
    if var_name not in source_ds.variables:
        raise ValueError("The Dataset source_ds does not contain the variable var_name!")
    fill_value = from_xarray(fill_value)
    if not isinstance(fill_value, (int, float, DataArray)):
        raise TypeError("The input fill_value must be of type int, float, or xr.DataArray!")
    if isinstance(fill_value, DataArray):
        da = source_ds[var_name]
        chan_shape = tuple(n for d, n in zip(da.dims, da.shape) if d != "channel")
        data = fill_value.data
        shape = tuple(n for n in data.shape if n != 1) if len(data.shape) != len(chan_shape) else tuple(data.shape)
        if shape != chan_shape:
            raise ValueError(f"If fill_value is an array it must be of the same shape as {var_name}!")
    return fill_value


def _mask_tensor(m, order):
    """uint8 device tensor of a mask with its dims in the order they appear in ``order``."""
    want = [d for d in order if d in m.dims]
    t = _dev(m)
    if list(m.dims) != want:
        t = t.permute([m.dims.index(d) for d in want])
    if t.dtype == torch.bool:  # the masks of echopype_amd.clean: reinterpreted, not copied
        return t.contiguous().view(torch.uint8)
    return (t != 0).to(torch.uint8).contiguous()


@xarray_io()
def apply_mask(source_ds, mask, var_name="Sv", fill_value=np.nan, storage_options_ds={},
               storage_options_mask={}):
    """Dataset like ``source_ds`` with ``var_name`` replaced by where(AND of masks, var, fill_value)."""
    _no_paths(source_ds, "source_ds")
    source_ds = from_xarray(source_ds)
    mask = _validate_and_collect_mask_input(mask, storage_options_mask)
    # (the reference checks var_name after the alignment, where a missing variable surfaces as a
    # KeyError from the dataset lookup; here the explicit messages come first)
    fill_value = _check_var_name_fill_value(source_ds, var_name, fill_value)
    source_ds = _check_mask_dim_alignment(source_ds, mask, var_name)

    source_da = source_ds[var_name]
    order = tuple(source_da.dims)
    src_t = _dev(source_da)
    if src_t.dtype not in (torch.float32, torch.float64):
        src_t = src_t.double()
    masks = mask if isinstance(mask, list) else [mask]
    # channel-carrying masks first so that the channel-less ones broadcast into them (xr.broadcast, :403)
    tensors = sorted((_mask_tensor(m, order) for m in masks), key=lambda t: -t.dim())
    final = tensors[0]  # (its shape is the shape of the AND of all: the others broadcast into it)
    for t in tensors[1:]:
        if final.numel() % t.numel() != 0 or tuple(final.shape[-t.dim():]) != tuple(t.shape):
            raise ValueError("All masks must have the same shape in the 'channel' dimension.")
    # up to four masks go to the kernel as they are (one sweep: the AND, the selection and the result's min / max); a
    # fifth and further ones are folded into the first beforehand
    while len(tensors) > 4:
        tensors = [ops.mask_and(tensors[0], tensors.pop())] + tensors[1:]
        final = tensors[0]
    has_chan = "channel" in order
    src_chan_shape = tuple(src_t.shape[1:]) if has_chan and order[0] == "channel" else tuple(
        n for d, n in zip(order, src_t.shape) if d != "channel")
    mask_has_chan = final.dim() == len(order) and has_chan
    mask_chan_shape = tuple(final.shape[1:]) if mask_has_chan else tuple(final.shape)
    if mask_chan_shape != src_chan_shape:
        raise ValueError(f"The final constructed mask is not of the same shape as source_ds[{var_name}] "
                         "along the ping_time, and range_sample dimensions!")
    if mask_has_chan and final.shape[0] != src_t.shape[0]:
        raise ValueError(f"If both the final constructed mask and source_ds[{var_name}] "
                         "have the channel dimension, that dimension should match between the two.")
    if has_chan and order[0] != "channel":
        raise NotImplementedError("the channel dimension of the source variable must come first")

    if isinstance(fill_value, DataArray):
        fill_t = _dev(fill_value, src_t.dtype).reshape(src_chan_shape).contiguous()
        out_t, mm = ops.apply_masks(src_t, tensors, fill_array=fill_t, want_minmax=True)
    else:
        out_t, mm = ops.apply_masks(src_t, tensors, fill_value=float(fill_value), want_minmax=True)

    output_ds = source_ds.copy()
    attrs = dict(source_da.attrs)
    lo, hi = mm.cpu().tolist()  # (a by-product of the sweep that wrote the array)
    attrs.update({
        "long_name": "Volume backscattering strength, masked (Sv re 1 m-1)",
        "actual_range": [round(lo, 2), round(hi, 2)],
        "history": f"{datetime.datetime.now(datetime.timezone.utc)}. `depth` calculated using:. "
                   "Created masked Sv dataarray.",
    })
    mattrs = dict(masks[0].attrs)
    if "history" in mattrs:
        attrs["history"] += f"\n{mattrs.pop('history')}"
    attrs.update(mattrs)
    output_ds[var_name] = DataArray(DeviceArray(out_t), order, attrs=attrs)
    prov = echopype_prov_attrs(process_type="mask")
    prov["mask_function"] = "mask.apply_mask"
    output_ds.attrs.update(prov)
    return insert_processing_level(output_ds, "L3*", input_ds=source_ds)
