#!/usr/bin/env python3
"""Generate tests/golden/ref_noise_goldens.npz by EXECUTING THE REFERENCE'S OWN
clean/api.py::estimate_background_noise / remove_background_noise (authoring container only).

Same approach as oracle/gen_chain_goldens.py: the function bodies (arithmetic on labelled arrays plus
``coarsen(boundary="pad").mean()``, ``min(dim)``, ``assign_coords`` and a forward-fill ``reindex``) run over
the strict named-dimension shim oracle/xr_shim.py; provenance / logging helpers are no-ops.
Output = data only: seeded inputs + the reference's Sv_noise and Sv_corrected.
"""
import logging
import os
import sys
import types

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import xr_shim  # noqa: E402
from gen_goldens import REF, _load  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "ref_noise_goldens.npz")
DA, DS = xr_shim.DataArray, xr_shim.Dataset


def load_clean_api():
    xr = types.ModuleType("xarray")
    xr.DataArray, xr.Dataset, xr.where, xr.merge, xr.apply_ufunc = DA, DS, xr_shim.where, xr_shim.merge, xr_shim.apply_ufunc
    sys.modules["xarray"] = xr
    for name in ("dask", "dask.array", "dask_image", "dask_image.ndfilters", "flox", "flox.xarray"):
        m = types.ModuleType(name)
        m.__path__ = []
        sys.modules[name] = m
    sys.modules["dask.array"].Array = np.ndarray
    sys.modules["dask"].array = sys.modules["dask.array"]
    for n, p in [("echopype", [REF]), ("echopype.clean", [f"{REF}/clean"]), ("echopype.utils", []),
                 ("echopype.commongrid", []), ("echopype.clean.transient_noise", [])]:
        m = types.ModuleType(n)
        m.__path__ = p
        sys.modules[n] = m
    cg = types.ModuleType("echopype.commongrid.utils")
    cg._convert_bins_to_interval_index = cg._parse_x_bin = None
    sys.modules[cg.__name__] = cg
    log = types.ModuleType("echopype.utils.log"); log._init_logger = logging.getLogger
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
    _load("echopype.clean.utils", f"{REF}/clean/utils.py")
    return _load("echopype.clean.api", f"{REF}/clean/api.py")


def case(api, g, tag, C, P, S, ping_num, rsn, nmax, snr, seed):
    rng = np.random.default_rng(seed)
    chans = np.array([f"ch{i}" for i in range(C)])
    pings = np.datetime64("2026-05-01T00:00:00", "ns") + np.arange(P) * np.timedelta64(1, "s")
    k = 0.19 * (1 + 0.3 * np.arange(C))[:, None, None] * (1 + 0.002 * rng.random((C, P, 1)))
    er = np.arange(S)[None, None, :] * k
    sv = -75 + 6 * rng.standard_normal((C, P, S)) + 20 * np.log10(np.maximum(er, 1)) * 0.3
    sv[rng.random((C, P, S)) < 0.02] = np.nan
    er[:, 2, S - 6:] = np.nan  # masked echo_range with its NaN Sv
    sv[:, 2, S - 6:] = np.nan
    sv[0, 5] = np.nan  # a whole ping missing
    alpha = np.array([0.0098, 0.0374, 0.0527])[:C, None] * (1 + 0.01 * rng.random((C, P)))
    dims = ["channel", "ping_time", "range_sample"]
    ds = DS(coords={"channel": chans, "ping_time": pings, "range_sample": np.arange(S)})
    ds["Sv"] = DA(sv, dims=dims)
    ds["echo_range"] = DA(er, dims=dims)
    ds["sound_absorption"] = DA(alpha, {"channel": chans, "ping_time": pings}, ["channel", "ping_time"])
    out = api.remove_background_noise(ds, ping_num, rsn, background_noise_max=nmax, SNR_threshold=snr)
    est = api.estimate_background_noise(ds, ping_num, rsn, background_noise_max=nmax)
    g[f"{tag}_Sv"], g[f"{tag}_echo_range"], g[f"{tag}_absorption"] = sv, er, alpha
    g[f"{tag}_args"] = np.array([ping_num, rsn, np.nan if nmax is None else float(nmax[:-2]), float(snr[:-2])])
    g[f"{tag}_Sv_noise"] = out["Sv_noise"].transpose(*dims).data
    g[f"{tag}_Sv_corrected"] = out["Sv_corrected"].transpose(*dims).data
    np.testing.assert_array_equal(est.transpose(*dims).data, g[f"{tag}_Sv_noise"])
    g[f"{tag}_noise_attrs_range"] = np.array(out["Sv_noise"].attrs["actual_range"] + out["Sv_corrected"].attrs["actual_range"])


def main():
    logging.disable(logging.WARNING)
    api = load_clean_api()
    g = {}
    case(api, g, "n0", 2, 23, 130, 5, 30, None, "3.0dB", 1)       # ragged last ping block / range block
    case(api, g, "n1", 3, 40, 100, 20, 50, "-125.0dB", "0.0dB", 2)
    case(api, g, "n2", 1, 7, 64, 10, 64, "-90.5dB", "5.5dB", 3)     # one ping block larger than the data
    np.savez_compressed(OUT, **g)
    print("wrote", os.path.normpath(OUT), f"{os.path.getsize(OUT)/1024:.1f} KiB,", len(g), "arrays")
    for t in ("n0", "n1", "n2"):
        c = g[f"{t}_Sv_corrected"]
        print(t, c.shape, "corrected kept:", int(np.isfinite(c).sum()), "of", c.size, "attrs", g[f"{t}_noise_attrs_range"])


if __name__ == "__main__":
    main()
