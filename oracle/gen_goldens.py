#!/usr/bin/env python3
"""Generate tests/golden/ref_leaf_goldens.npz by EXECUTING THE REFERENCE'S OWN leaf functions.

Run only in the authoring container (needs /root/reference); the output is a small data
fixture (inputs + expected outputs) that travels to the GPU box, the reference does not.

The reference package cannot be imported (xarray, dask, flox, zarr ... are missing and it
needs Python >= 3.11), but its pure numpy/scipy leaf functions run when loaded by file
path with stub modules standing in for ``xarray`` and ``echopype.convert.set_groups_ek80``
(three string constants, convert/set_groups_ek80.py:16-18)  -- SURVEY.md Appendix B.

Functions executed:
  utils/uwa.py::calc_sound_speed, calc_absorption
  calibrate/ek80_complex.py::tapered_chirp, filter_decimate_chirp, _convolve_per_channel,
                             get_tau_effective, get_norm_fac
"""
import importlib.util
import os
import sys
import types

import numpy as np

REF = "/root/reference/echopype"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden",
                   "ref_leaf_goldens.npz")


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def load_reference_leafs():
    uwa = _load("ref_uwa", f"{REF}/utils/uwa.py")
    xr = types.ModuleType("xarray")

    class _DA:  # enough for get_tau_effective / get_norm_fac return values
        def __init__(self, data=None, coords=None, dims=None, **kw):
            self.data = np.asarray(data)
            self.coords = coords

    xr.DataArray = _DA
    xr.Dataset = object
    xr.where = np.where
    xr.apply_ufunc = None
    sys.modules["xarray"] = xr
    for n, p in [("echopype", [REF]), ("echopype.calibrate", [f"{REF}/calibrate"]),
                 ("echopype.convert", [])]:
        m = types.ModuleType(n)
        m.__path__ = p
        sys.modules[n] = m
    sg = types.ModuleType("echopype.convert.set_groups_ek80")
    sg.DECIMATION, sg.FILTER_IMAG, sg.FILTER_REAL = "deci_fac", "coeffs_imag", "coeffs_real"
    sys.modules["echopype.convert.set_groups_ek80"] = sg
    ek = _load("echopype.calibrate.ek80_complex", f"{REF}/calibrate/ek80_complex.py")
    return uwa, ek


class _Label:
    def __init__(self, v):
        self.values = v


def main():
    uwa, ek = load_reference_leafs()
    g = {}

    # ---- uwa: the points of tests/utils/test_utils_uwa.py:15-22 plus a dense sweep
    pts = np.array([
        [18000, 27, 35, 10, 8], [18000, 27, 35, 100, 8], [38000, 27, 35, 10, 8],
        [38000, 10, 35, 10, 8], [120000, 27, 35, 10, 8], [200000, 27, 35, 10, 8],
        [455000, 20, 35, 10, 8], [1000000, 10, 35, 10, 8],
        [70000, 4.5, 33.2, 250, 7.9], [333000, 19.9, 30.0, 5, 8.2], [38000, 20.0, 35, 10, 8],
    ], dtype=np.float64)
    g["uwa_pts"] = pts
    for src in ("AM", "FG", "AZFP"):
        g[f"uwa_abs_{src}"] = np.array([
            uwa.calc_absorption(frequency=f, temperature=T, salinity=S, pressure=P, pH=pH,
                                formula_source=src) for f, T, S, P, pH in pts])
    g["uwa_abs_FG_c1500"] = np.array([
        uwa.calc_absorption(frequency=f, temperature=T, salinity=S, pressure=P, pH=pH,
                            sound_speed=1500.0, formula_source="FG") for f, T, S, P, pH in pts])
    ss_pts = np.array([[27, 35, 10], [27, 35, 100], [5, 35, 3500], [-1.5, 30.5, 7000],
                       [12.25, 38.0, 0]], dtype=np.float64)
    g["uwa_ss_pts"] = ss_pts
    for src in ("Mackenzie", "AZFP"):
        g[f"uwa_ss_{src}"] = np.array([
            uwa.calc_sound_speed(temperature=T, salinity=S, pressure=P, formula_source=src)
            for T, S, P in ss_pts])
    # vector-frequency form used by the calibrators (frequency is an array)
    fvec = np.array([18e3, 38e3, 70e3, 120e3, 200e3, 333e3])
    g["uwa_fvec"] = fvec
    g["uwa_abs_FG_vec"] = uwa.calc_absorption(frequency=fvec, temperature=10, salinity=35,
                                              pressure=10, pH=8, formula_source="FG")

    # ---- transmit replica: two channels, deterministic "vendor" filters
    k47, k91 = np.arange(47), np.arange(91)
    wbt = (np.hanning(47) * np.exp(2j * np.pi * 0.045 * k47) / 10).astype(np.complex64)
    pc = (np.hanning(91) * np.exp(2j * np.pi * 0.13 * k91) / 20).astype(np.complex64)
    g["wbt_fil"], g["pc_fil"] = wbt, pc
    chans = [
        dict(fs=1.5e6, tau=1.024e-3, slope=0.05, f0=45e3, f1=90e3),     # 70 kHz BB
        dict(fs=1.5e6, tau=0.512e-3, slope=0.05, f0=90e3, f1=170e3),    # 120 kHz BB
        dict(fs=1.5e6, tau=1.024e-3, slope=0.05, f0=38e3, f1=38e3),     # CW replica
    ]
    g["chan_params"] = np.array([[c["fs"], c["tau"], c["slope"], c["f0"], c["f1"]] for c in chans])
    replicas = []
    for i, c in enumerate(chans):
        for drop in (False, True):
            y, t = ek.tapered_chirp(c["fs"], np.array([c["tau"]]), np.array([c["slope"]]),
                                    np.array([c["f0"]]), np.array([c["f1"]]),
                                    drop_last_hanning_zero=drop)
            g[f"chirp{i}_drop{int(drop)}"] = y
        y, t = ek.tapered_chirp(c["fs"], np.array([c["tau"]]), np.array([c["slope"]]),
                                np.array([c["f0"]]), np.array([c["f1"]]))
        coeff = dict(wbt_fil=wbt, wbt_decifac=6, pc_fil=pc, pc_decifac=2)
        ytx, ttx = ek.filter_decimate_chirp(coeff, y, c["fs"])
        g[f"replica{i}"], g[f"replica{i}_time"] = ytx, ttx
        replicas.append(ytx)
        fs_deci = 1 / np.diff(ttx[:2])
        mode = "CW" if c["f0"] == c["f1"] else "BB"
        te = ek.get_tau_effective({"ch": ytx}, {"ch": fs_deci}, mode, None, None).data
        g[f"tau_eff{i}"] = np.asarray(te, dtype=np.float64).reshape(-1)
        g[f"norm_fac{i}"] = np.asarray(ek.get_norm_fac({"ch": ytx}).data, dtype=np.float64)

    # ---- matched filter: _convolve_per_channel on a seeded (S, C=2) slab with echoes,
    #      one NaN-zeroed tail, plus an all-zero slab
    rng = np.random.default_rng(20260501)
    S = 700
    slab = (rng.standard_normal((S, 2)) + 1j * rng.standard_normal((S, 2))) * 1e-3
    for ch in range(2):
        r = replicas[ch]
        for start in (50 + 37 * ch, 300, 610):   # last one is cut by the end of the ping
            n = min(r.size, S - start)
            slab[start:start + n, ch] += 0.5 * r[:n]
    slab[-33:, :] = 0.0   # what compress_pulse hands over after NaN -> 0
    g["mf_slab_in"] = slab
    rep_dict = {"c0": np.flipud(np.conj(replicas[0])), "c1": np.flipud(np.conj(replicas[1]))}
    out = ek._convolve_per_channel(slab.copy(), rep_dict, [_Label("c0"), _Label("c1")])
    assert out.dtype == np.complex64
    g["mf_slab_out"] = out
    zero = np.zeros((S, 2), dtype=np.complex128)
    g["mf_zero_out_is_input"] = np.array(
        ek._convolve_per_channel(zero, rep_dict, [_Label("c0"), _Label("c1")]) is zero)

    np.savez_compressed(OUT, **g)
    print("wrote", os.path.normpath(OUT), f"{os.path.getsize(OUT)/1024:.1f} KiB,", len(g), "arrays")


if __name__ == "__main__":
    main()
