#!/usr/bin/env python3
"""Generate tests/golden/ref_mask_goldens.npz by EXECUTING THE REFERENCE'S OWN leaf functions
for the noise masks (authoring container only, needs /root/reference).

clean/utils.py imports xarray / flox / dask_image at module level (all absent here); its two
single-channel numpy functions need none of them, so the module is loaded by file path with empty
stand-in modules for those imports.  Functions executed:
  clean/utils.py::echopy_impulse_noise_mask, echopy_attenuated_signal_mask
  utils/compute.py::_log2lin, _lin2log
The output is data only (seeded inputs + the reference's outputs).
"""
import os
import sys
import types

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_goldens import REF, _load  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden",
                   "ref_mask_goldens.npz")


def load_clean_utils():
    for name in ("dask", "dask.array", "dask_image", "dask_image.ndfilters", "flox", "flox.xarray",
                 "xarray"):
        m = types.ModuleType(name)
        m.__path__ = []
        sys.modules[name] = m
    sys.modules["dask.array"].Array = np.ndarray
    sys.modules["dask"].array = sys.modules["dask.array"]
    sys.modules["xarray"].Dataset = sys.modules["xarray"].DataArray = object
    for n, p in [("echopype", [REF]), ("echopype.clean", [f"{REF}/clean"]),
                 ("echopype.utils", []), ("echopype.commongrid", [])]:
        m = types.ModuleType(n)
        m.__path__ = p
        sys.modules[n] = m
    cg = types.ModuleType("echopype.commongrid.utils")
    cg._convert_bins_to_interval_index = None  # only used by the flox helper, not executed
    sys.modules["echopype.commongrid.utils"] = cg
    _load("echopype.utils.compute", f"{REF}/utils/compute.py")
    return _load("echopype.clean.utils", f"{REF}/clean/utils.py")


def main():
    cu = load_clean_utils()
    rng = np.random.default_rng(20260927)
    g = {}

    # ---- impulse: (range_sample, ping_time) blocks with spikes, NaNs and +/-inf
    for i, (S, P, n, thr) in enumerate([(12, 40, 2, 10.0), (7, 25, 1, 6.0), (5, 9, 4, 3.0), (3, 30, 5, 10.0)]):
        sv = -70 + 3 * rng.standard_normal((S, P))
        sv[rng.random((S, P)) < 0.08] += 25.0          # impulses
        sv[rng.random((S, P)) < 0.05] = np.nan
        if i == 1:
            sv[2, 5], sv[3, 11] = -np.inf, np.inf
        g[f"imp{i}_sv"], g[f"imp{i}_args"] = sv, np.array([n, thr])
        g[f"imp{i}_mask"] = cu.echopy_impulse_noise_mask(sv.copy(), n, thr)

    # ---- attenuated signal: (ping_time, range_sample); ranges with per-ping jitter, NaN tails,
    #      attenuated pings, all-NaN layer, -inf samples
    for i, (P, S, n, thr, up, lw) in enumerate([(60, 80, 5, -6.0, 20.0, 45.0), (33, 50, 3, -4.0, 5.0, 30.0),
                                                (20, 40, 15, -5.0, 10.0, 20.0), (45, 64, 2, -3.0, 0.0, 1e9)]):
        rg = np.arange(S)[None, :] * (0.75 + 0.01 * rng.random((P, 1))) + 0.2 * rng.random((P, 1))
        sv = -65 + 4 * rng.standard_normal((P, S))
        att = rng.random(P) < 0.15
        sv[att] -= 12.0
        sv[rng.random((P, S)) < 0.06] = np.nan
        if i == 0:
            sv[17, :] = np.nan                           # whole ping NaN -> never masked
            sv[30, 3:9] = -np.inf
        if i == 1:
            rg[::3, 44:] = np.nan                        # NaN range tail: argmin picks the first NaN
            sv[::3, 44:] = np.nan
        g[f"att{i}_sv"], g[f"att{i}_range"] = sv, rg
        g[f"att{i}_args"] = np.array([up, lw, n, thr])
        g[f"att{i}_mask"] = cu.echopy_attenuated_signal_mask(sv.copy(), rg.copy(), up, lw, n, thr)
        assert g[f"att{i}_mask"].any() or i == 2, i

    x = np.array([-120.0, -70.5, 0.0, 3.0, np.nan, -np.inf])
    g["db_in"], g["log2lin"] = x, sys.modules["echopype.utils.compute"]._log2lin(x)
    with np.errstate(divide="ignore"):
        g["lin2log"] = sys.modules["echopype.utils.compute"]._lin2log(g["log2lin"])

    np.savez_compressed(OUT, **g)
    print("wrote", os.path.normpath(OUT), f"{os.path.getsize(OUT)/1024:.1f} KiB,", len(g), "arrays")
    for k in g:
        if k.endswith("_mask"):
            print(k, g[k].shape, int(g[k].sum()))


if __name__ == "__main__":
    main()
