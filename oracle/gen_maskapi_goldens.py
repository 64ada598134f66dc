#!/usr/bin/env python3
"""Generate tests/golden/ref_maskapi_goldens.npz by EXECUTING THE REFERENCE'S OWN
clean/api.py::mask_transient_noise / mask_impulse_noise / mask_attenuated_signal end to end
(authoring container only, needs /root/reference).

oracle/gen_mask_goldens.py pins the two single-channel numpy leaves; this script pins everything AROUND
them: the dB / metre string parsing (commongrid/utils.py::_parse_x_bin, clean/utils.py::extract_dB), the
per-channel samples-per-bin rule, coarsen + forward-fill up-sampling, the (2n+1)x(2m+1) reflect pooling
below ``exclude_above`` with its flattened-argmin start index, the value-window triple loop of pool_Sv,
the early return of the attenuated-signal mask and the dimension order each function returns.

The function bodies run over the strict named-dimension shim oracle/xr_shim.py (xarray is not installable
here).  dask_image's generic_filter is a chunked wrapper around scipy.ndimage.generic_filter; the stand-in
calls scipy's directly (windows kept no larger than the array, where scipy's reflect mode is well defined).
The flox-based ``downsample_upsample_along_depth`` (use_index_binning=False for the impulse mask) cannot run
without flox and stays pinned by the restated known-answer tests only.
Output = data only: seeded inputs, the arguments and the reference's masks.
"""
import logging
import os
import sys
import types

import numpy as np
import scipy.ndimage

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import xr_shim  # noqa: E402
from gen_goldens import REF, _load  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "ref_maskapi_goldens.npz")
DA, DS = xr_shim.DataArray, xr_shim.Dataset
DIMS = ["channel", "ping_time", "range_sample"]


class _Lazy:
    def __init__(self, a):
        self.a = a

    def compute(self):
        return self.a


def load_clean_api():
    xr = types.ModuleType("xarray")
    for n in ("DataArray", "Dataset", "where", "merge", "apply_ufunc", "concat", "full_like", "zeros_like"):
        setattr(xr, n, getattr(xr_shim, n))
    sys.modules["xarray"] = xr
    for name in ("dask", "dask.array", "dask_image", "dask_image.ndfilters", "flox", "flox.xarray", "geopy"):
        m = types.ModuleType(name)
        m.__path__ = []
        sys.modules[name] = m
    sys.modules["dask.array"].Array = np.ndarray
    sys.modules["dask"].array = sys.modules["dask.array"]
    sys.modules["flox.xarray"].xarray_reduce = None  # only the flox paths use it; they are not executed
    sys.modules["geopy"].distance = None
    nd = sys.modules["dask_image.ndfilters"]
    nd.generic_filter = lambda a, function, size, mode: _Lazy(
        scipy.ndimage.generic_filter(np.asarray(a, dtype=float), function, size=size, mode=mode))
    sys.modules["dask_image"].ndfilters = nd
    for n, p in [("echopype", [REF]), ("echopype.clean", [f"{REF}/clean"]), ("echopype.utils", []),
                 ("echopype.commongrid", []), ("echopype.consolidate", []), ("echopype.clean.transient_noise", [])]:
        m = types.ModuleType(n)
        m.__path__ = p
        sys.modules[n] = m
    ca = types.ModuleType("echopype.consolidate.api")
    ca.POSITION_VARIABLES = ["latitude", "longitude"]
    sys.modules[ca.__name__] = ca
    log = types.ModuleType("echopype.utils.log")
    log._init_logger = logging.getLogger
    sys.modules[log.__name__] = log
    prov = types.ModuleType("echopype.utils.prov")
    prov.add_processing_level = lambda *a, **k: (lambda f: f)
    prov.echopype_prov_attrs = lambda process_type="processing": {}
    prov.insert_input_processing_level = lambda ds, input_ds=None: ds
    sys.modules[prov.__name__] = prov
    for n, attr in (("transient_fielding", "transient_noise_fielding"), ("transient_matecho", "transient_noise_matecho")):
        m = types.ModuleType(f"echopype.clean.transient_noise.{n}")
        setattr(m, attr, None)
        sys.modules[m.__name__] = m
    _load("echopype.utils.compute", f"{REF}/utils/compute.py")
    _load("echopype.commongrid.utils", f"{REF}/commongrid/utils.py")  # the real _parse_x_bin
    _load("echopype.clean.utils", f"{REF}/clean/utils.py")
    return _load("echopype.clean.api", f"{REF}/clean/api.py")


def synth(C, P, S, seed, step=0.19):
    rng = np.random.default_rng(seed)
    chans = np.array([f"ch{i}" for i in range(C)])
    pings = np.datetime64("2026-05-01T00:00:00", "ns") + np.arange(P) * np.timedelta64(1, "s")
    k = step * (1 + 0.3 * np.arange(C))[:, None, None] * np.ones((C, P, 1))
    er = np.arange(S)[None, None, :] * k
    sv = -72 + 4 * rng.standard_normal((C, P, S))
    sv[rng.random((C, P, S)) < 0.04] += 22.0                   # spikes
    sv[:, rng.random(P) < 0.12, :] += 14.0                     # loud pings (transient)
    sv[:, rng.random(P) < 0.10, :] -= 13.0                     # weak pings (attenuated)
    sv[rng.random((C, P, S)) < 0.03] = np.nan
    ds = DS(coords={"channel": chans, "ping_time": pings, "range_sample": np.arange(S)})
    ds["Sv"] = DA(sv, dims=DIMS)
    ds["echo_range"] = DA(er, dims=DIMS)
    return ds, sv, er


def main():
    api = load_clean_api()
    g = {}

    def put(tag, sv, er, mask, dims, **kw):
        g[f"{tag}_Sv"], g[f"{tag}_echo_range"] = sv, er
        g[f"{tag}_mask"] = np.asarray(mask.data)
        g[f"{tag}_dims"] = np.array(list(dims))
        g[f"{tag}_kw"] = np.array([f"{k}={v!r}" for k, v in kw.items()])

    # ---- transient noise, index binning (mean and median pooling)
    for i, (C, P, S, kw) in enumerate([
        (2, 30, 60, dict(func="nanmean", depth_bin="2m", num_side_pings=3, exclude_above="1.5m",
                         transient_noise_threshold="6.0dB")),
        (2, 24, 48, dict(func="nanmedian", depth_bin="1m", num_side_pings=2, exclude_above="3.0m",
                         transient_noise_threshold="4.5dB")),
        (3, 12, 40, dict(func="nanmean", depth_bin="0.5m", num_side_pings=5, exclude_above="0.0m",
                         transient_noise_threshold="12.0dB")),
    ]):
        ds, sv, er = synth(C, P, S, 100 + i)
        m = api.mask_transient_noise(ds, range_var="echo_range", use_index_binning=True,
                                     chunk_dict={"ping_time": 8, "range_sample": 16}, **kw)
        put(f"tri{i}", sv, er, m, m.dims, **kw)

    # ---- transient noise, value windows (the triple loop of pool_Sv)
    for i, (C, P, S, kw) in enumerate([
        (1, 9, 16, dict(func="nanmean", depth_bin="0.5m", num_side_pings=2, exclude_above="0.3m",
                        transient_noise_threshold="5.0dB")),
        (2, 7, 12, dict(func="nanmedian", depth_bin="0.4m", num_side_pings=1, exclude_above="0.0m",
                        transient_noise_threshold="3.0dB")),
    ]):
        ds, sv, er = synth(C, P, S, 200 + i)
        m = api.mask_transient_noise(ds, range_var="echo_range", use_index_binning=False, **kw)
        put(f"trv{i}", sv, er, m, m.dims, **kw)

    # ---- impulse noise, index binning
    for i, (C, P, S, kw) in enumerate([
        (2, 30, 50, dict(depth_bin="1m", num_side_pings=2, impulse_noise_threshold="10.0dB")),
        (3, 16, 33, dict(depth_bin="0.7m", num_side_pings=1, impulse_noise_threshold="6.0dB")),
        (1, 20, 10, dict(depth_bin="5m", num_side_pings=3, impulse_noise_threshold="8.0dB")),  # one bin > S
    ]):
        ds, sv, er = synth(C, P, S, 300 + i)
        m = api.mask_impulse_noise(ds, range_var="echo_range", use_index_binning=True, **kw)
        put(f"imp{i}", sv, er, m, m.dims, **kw)

    # ---- attenuated signal (incl. the empty-mask early returns)
    for i, (C, P, S, kw) in enumerate([
        (2, 40, 60, dict(upper_limit_sl="3.0m", lower_limit_sl="8.0m", num_side_pings=4,
                         attenuation_signal_threshold="8.0dB")),
        (3, 25, 45, dict(upper_limit_sl="1.0m", lower_limit_sl="2.5m", num_side_pings=2,
                         attenuation_signal_threshold="5.0dB")),
        (2, 12, 30, dict(upper_limit_sl="30.0m", lower_limit_sl="40.0m", num_side_pings=2,
                         attenuation_signal_threshold="8.0dB")),  # search range below the echogram
    ]):
        ds, sv, er = synth(C, P, S, 400 + i)
        m = api.mask_attenuated_signal(ds, range_var="echo_range", **kw)
        put(f"att{i}", sv, er, m, m.dims, **kw)

    np.savez_compressed(OUT, **g)
    for k in sorted(g):
        if k.endswith("_mask"):
            print(k, g[k].shape, g[k].dtype, int(np.sum(g[k])), "of", g[k].size, "dims", list(g[k[:-5] + "_dims"]))
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
