#!/usr/bin/env python3
"""Generate tests/golden/ref_mvbs_index_goldens.npz by EXECUTING THE REFERENCE'S OWN
commongrid/api.py::compute_MVBS_index_binning and the host-side parsing helpers of commongrid/utils.py
(authoring container only, needs /root/reference).

compute_MVBS_index_binning is plain labelled-array arithmetic (10**(Sv/10), a joint
``coarsen(ping_time, range_sample, boundary="pad").mean(skipna=True)``, ``.min`` of echo_range) and runs over
the strict shim oracle/xr_shim.py; the coordinates of the coarsened dimensions follow xarray's default
``coord_func="mean"`` (documented behaviour of DataArray.coarsen), restated in the shim.
compute_MVBS / compute_NASC themselves need flox and cannot run here; what CAN run are the reference's
own argument parsers (``_parse_x_bin``, ``ping_time_bin_parsing_and_conversion``), executed on a table of
accepted and rejected strings -- the accepted values and the exception type + message are the golden.
Output = data only.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import xr_shim  # noqa: E402
from gen_goldens import REF, _load  # noqa: E402
from gen_maskapi_goldens import load_clean_api  # noqa: E402  (sets up the stub modules)

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "ref_mvbs_index_goldens.npz")
DA, DS = xr_shim.DataArray, xr_shim.Dataset
DIMS = ["channel", "ping_time", "range_sample"]


def main():
    load_clean_api()
    cg_api = _load("echopype.commongrid.api", f"{REF}/commongrid/api.py")
    cg_utils = sys.modules["echopype.commongrid.utils"]
    g = {}
    for i, (C, P, S, pn, rn, dt_ns) in enumerate([(2, 20, 50, 3, 7, 1_000_000_000), (3, 9, 16, 4, 4, 333_333_333),
                                                  (1, 7, 5, 10, 2, 2_500_000_001), (2, 12, 30, 1, 30, 1_000_000_000)]):
        rng = np.random.default_rng(500 + i)
        chans = np.array([f"ch{k}" for k in range(C)])
        jitter = rng.integers(0, 1000, P)
        pings = np.datetime64("2026-05-01T00:00:00", "ns") + (np.arange(P) * dt_ns + jitter).astype("timedelta64[ns]")
        er = np.arange(S)[None, None, :] * (0.19 * (1 + 0.3 * np.arange(C)))[:, None, None] * np.ones((C, P, 1))
        sv = -70 + 5 * rng.standard_normal((C, P, S))
        sv[rng.random((C, P, S)) < 0.1] = np.nan
        sv[0, :min(pn, P), :rn] = np.nan  # one all-NaN block
        er[:, 1, S - 3:] = np.nan
        ds = DS(coords={"channel": chans, "ping_time": pings, "range_sample": np.arange(S)})
        ds["Sv"], ds["echo_range"] = DA(sv, dims=DIMS), DA(er, dims=DIMS)
        ds["frequency_nominal"] = DA(np.arange(C) * 1e4 + 38e3, {"channel": chans}, ["channel"])
        out = cg_api.compute_MVBS_index_binning(ds, range_sample_num=rn, ping_num=pn)
        t = f"ix{i}"
        g[f"{t}_Sv"], g[f"{t}_echo_range"], g[f"{t}_ping_time"], g[f"{t}_args"] = sv, er, pings, np.array([rn, pn])
        g[f"{t}_out_Sv"] = out["Sv"].transpose(*DIMS).data
        g[f"{t}_out_echo_range"] = out["echo_range"].transpose(*DIMS).data
        g[f"{t}_out_ping_time"] = np.asarray(out["ping_time"].data)
        g[f"{t}_out_range_sample"] = np.asarray(out["range_sample"].data)
        print(t, g[f"{t}_out_Sv"].shape, g[f"{t}_out_ping_time"][:2])

    # ---- the reference's argument parsers on accepted / rejected strings
    rows = []
    for label, vals in (("range_bin", ["10m", "0.5m", "5 m", "20.0m", ".5m", "1,5m", "10", "10km", "m", "10nmi", 5, "abc"]),
                        ("dist_bin", ["0.5nmi", "1nmi", "2 nmi", "10m", "nmi", "0.5", 1.0])):
        for v in vals:
            try:
                rows.append((label, repr(v), "ok", repr(float(cg_utils._parse_x_bin(v, label)))))
            except Exception as e:  # noqa: BLE001
                rows.append((label, repr(v), type(e).__name__, str(e)))
    for v in ["20s", "10S", "1min", "2T", "1h", "1H", "0.5s", "500ms", "1D", "20", "s", 20, "1.5min", "3Min"]:
        try:
            rows.append(("ping_time_bin", repr(v), "ok", repr(cg_utils.ping_time_bin_parsing_and_conversion(v))))
        except Exception as e:  # noqa: BLE001
            rows.append(("ping_time_bin", repr(v), type(e).__name__, str(e)))
    g["parse_rows"] = np.array(rows)
    for r in rows:
        print(r)
    np.savez_compressed(OUT, **g)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
