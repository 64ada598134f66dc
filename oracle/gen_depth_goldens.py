#!/usr/bin/env python3
"""Generate tests/golden/ref_depth_goldens.npz by EXECUTING THE REFERENCE'S OWN
consolidate/ek_depth_utils.py (authoring container only, needs /root/reference): the inputs add_depth derives
from an EK EchoData -- transducer depth from the Platform vertical offsets, echo-range scaling from platform
pitch / roll (scipy Rotation, as the reference evaluates it) and from the beam direction vectors -- and
utils/align.py::align_to_ping_time in the branches that need no interpolation (identical time axis, a single
time).  The nearest-neighbour interpolation branch needs xarray.interp and stays pinned by the reference's
known-answer tests restated in tests/test_host_logic.py.  Output = data only.
"""
import logging
import os
import sys
import types

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import xr_shim  # noqa: E402
from gen_goldens import REF, _load  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "ref_depth_goldens.npz")
DA, DS = xr_shim.DataArray, xr_shim.Dataset


def load():
    xr = types.ModuleType("xarray")
    for n in ("DataArray", "Dataset", "where"):
        setattr(xr, n, getattr(xr_shim, n))
    sys.modules["xarray"] = xr
    for n, p in [("echopype", [REF]), ("echopype.consolidate", [f"{REF}/consolidate"]), ("echopype.utils", [f"{REF}/utils"])]:
        m = types.ModuleType(n)
        m.__path__ = p
        sys.modules[n] = m
    log = types.ModuleType("echopype.utils.log")
    log._init_logger = logging.getLogger
    sys.modules[log.__name__] = log
    _load("echopype.utils.align", f"{REF}/utils/align.py")
    return _load("echopype.consolidate.ek_depth_utils", f"{REF}/consolidate/ek_depth_utils.py")


def main():
    eku = load()
    logging.disable(logging.WARNING)
    rng = np.random.default_rng(20260928)
    g = {}
    P = 9
    pt = np.datetime64("2026-05-01T00:00:00", "ns") + np.arange(P) * np.timedelta64(3, "s")
    ping_time = DA(pt, {"ping_time": pt}, ["ping_time"], "ping_time")

    # ---- vertical offsets: time2 identical to ping_time, and a single time2
    for tag, t2 in (("same", pt), ("single", pt[:1])):
        n = len(t2)
        plat = DS(coords={"time2": t2})
        vals = {k: rng.uniform(-3, 6, n) for k in ("water_level", "vertical_offset", "transducer_offset_z")}
        if tag == "same":
            vals["vertical_offset"][4] = np.nan
        for k, v in vals.items():
            plat[k] = DA(v, {"time2": t2}, ["time2"], k)
            g[f"vo_{tag}_{k}"] = v
        out = eku.ek_use_platform_vertical_offsets(plat, ping_time)
        assert out.dims == ("ping_time",)
        g[f"vo_{tag}_time2"], g[f"vo_{tag}_out"] = t2, np.asarray(out.data, dtype=np.float64)

    # ---- platform angles (scipy Rotation from_euler("ZYX"), element [2, 2])
    for tag, t2 in (("same", pt), ("single", pt[:1])):
        n = len(t2)
        plat = DS(coords={"time2": t2})
        pitch, roll = rng.uniform(-25, 25, n), rng.uniform(-40, 40, n)
        if tag == "same":
            pitch[2], roll[6] = np.nan, 0.0
        plat["pitch"], plat["roll"] = DA(pitch, {"time2": t2}, ["time2"], "pitch"), DA(roll, {"time2": t2}, ["time2"], "roll")
        out = eku.ek_use_platform_angles(plat, ping_time)
        g[f"pa_{tag}_pitch"], g[f"pa_{tag}_roll"], g[f"pa_{tag}_time2"] = pitch, roll, t2
        g[f"pa_{tag}_out"] = np.asarray(out.data, dtype=np.float64)

    # ---- beam angles: normalised, not normalised, zero and NaN vectors
    ch = np.array([f"ch{i}" for i in range(6)])
    v = rng.standard_normal((6, 3))
    v[0] /= np.linalg.norm(v[0])          # normalised
    v[1] = [0.0, 0.0, 1.0]
    v[2] *= 3.0                           # not normalised
    v[3] = 0.0                            # zero vector -> NaN
    v[4] = [1e-9, 0.0, 0.0]               # below the tolerance -> NaN
    v[5, 1] = np.nan
    beam = DS(coords={"channel": ch})
    for i, k in enumerate(("beam_direction_x", "beam_direction_y", "beam_direction_z")):
        beam[k] = DA(v[:, i], {"channel": ch}, ["channel"], k)
    out = eku.ek_use_beam_angles(beam)
    g["ba_vectors"], g["ba_out"] = v, np.asarray(out.data, dtype=np.float64)

    np.savez_compressed(OUT, **g)
    for k in sorted(g):
        if k.endswith("_out"):
            print(k, np.round(g[k], 6))
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
