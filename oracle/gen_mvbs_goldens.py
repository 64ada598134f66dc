#!/usr/bin/env python3
"""Generate tests/golden/ref_mvbs_goldens.npz by EXECUTING THE REFERENCE'S OWN brute-force expectations of
compute_MVBS and compute_NASC (authoring container only, needs /root/reference).

compute_MVBS / compute_NASC themselves are flox group-bys (commongrid/utils.py:504-628, :97-205) and cannot run
here (no xarray / flox).  What the reference's test-suite holds them to CAN: triple Python loops over bins that
select by label and average with plain NumPy --
  /root/reference/echopype/tests/mock_data.py:28-85            _get_expected_mvbs_val
  /root/reference/echopype/tests/commongrid/conftest.py:466-546  _get_expected_nasc_val_nanmean
  /root/reference/echopype/tests/commongrid/conftest.py:548-617  _brute_nanmean_reduce_3d
  /root/reference/echopype/tests/commongrid/conftest.py:405-447  _create_dataset, get_NASC_echoview
(test_commongrid_api.py:371-436, :447-470 assert compute_MVBS / compute_NASC == these, atol = rtol = 1e-10).
Both files are loaded BY PATH and their functions executed on datasets made by the reference's own generators
(mock_data.py:88-214, seeded) over oracle/xr_shim.py.  The shim supplies labelled-array plumbing only:
``.sel(dim=slice(a, b))`` and ``.resample(...).first().indexes`` are delegated to pandas (Index.slice_indexer,
Series.resample -- what xarray itself calls), ``.max()`` is NumPy's nanmax.  ``get_distance_from_latlon`` needs
geopy (absent): the cumulative distance per ping is an INPUT here (seeded positive steps), injected in place of
that call; the array pass under test starts from it.

The brute-force selects pings with label slices that are inclusive at BOTH ends, so it equals the left-closed
group-by only when no ping time / distance sits exactly on a bin edge; the script asserts that for every case
(as holds for the reference's own fixtures).
Output = data only (inputs + expected outputs).
"""
import os
import sys
import types

import numpy as np
import pandas as pd

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import xr_shim  # noqa: E402
from gen_goldens import REF, _load  # noqa: E402
from gen_maskapi_goldens import load_clean_api  # noqa: E402  (sets up the stub modules)

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "ref_mvbs_goldens.npz")
DIMS = ["channel", "ping_time", "range_sample"]


def _raw(fixture):
    """The plain function behind a @pytest.fixture (pytest refuses to call fixtures directly)."""
    for attr in ("__wrapped__", "_fixture_function"):
        if hasattr(fixture, attr):
            return getattr(fixture, attr)
    return fixture.__pytest_wrapped__.obj


def load_reference_expectations():
    load_clean_api()
    t = types.ModuleType("echopype.tests")
    t.__path__ = []
    sys.modules["echopype.tests"] = t
    mock = _load("echopype.tests.mock_data", f"{REF}/tests/mock_data.py")
    ep = sys.modules["echopype"]
    ep.utils = sys.modules["echopype.utils"]
    ep.utils.compute = sys.modules["echopype.utils.compute"]
    cons = sys.modules["echopype.consolidate"]

    def _no_add_depth(*a, **k):  # the fixtures that call it are not executed (depth is an input here)
        raise RuntimeError("add_depth is not available in the authoring container")

    cons.add_depth = _no_add_depth
    conf = _load("ref_commongrid_conftest", f"{REF}/tests/commongrid/conftest.py")
    return mock, conf


def _no_edge_hits(values, edges):
    assert not np.isin(np.asarray(values), np.asarray(edges)).any(), "a ping sits exactly on a bin edge"


def main():
    mock, conf = load_reference_expectations()
    g = {}
    nan_ilocs = _raw(conf.mock_nan_ilocs)()
    params = _raw(conf.mock_parameters)()
    sample = _raw(conf.mock_Sv_sample)(params)

    def add(tag, ds, ping_time_bin, range_bin, dist_steps, dist_bin, nasc_range_bin, depth):
        """Execute both expectations on ``ds`` and record inputs + outputs."""
        C = ds["Sv"].shape[0]
        # --- MVBS
        idx = ds["ping_time"].resample(ping_time=ping_time_bin, skipna=True).first().indexes["ping_time"]
        edges = idx.union([idx[-1] + pd.Timedelta(ping_time_bin)]).values
        _no_edge_hits(ds["ping_time"].data[1:], edges)  # the very first ping may open the first bin
        exp = mock._get_expected_mvbs_val(ds, ping_time_bin, range_bin, C)
        g[f"{tag}_Sv"], g[f"{tag}_echo_range"] = ds["Sv"].data.copy(), ds["echo_range"].data.copy()
        g[f"{tag}_ping_time"] = np.asarray(ds["ping_time"].data).astype("datetime64[ns]")
        g[f"{tag}_ping_time_bin"], g[f"{tag}_range_bin"] = np.array(ping_time_bin), np.array(float(range_bin))
        g[f"{tag}_mvbs"] = np.asarray(exp)
        g[f"{tag}_mvbs_time_labels"] = np.asarray(idx.values).astype("datetime64[ns]")
        # --- NASC (depth and the cumulative distance are inputs)
        ds["depth"] = xr_shim.DataArray(depth, dims=DIMS)
        dist = np.cumsum(dist_steps)
        d_edges = np.arange(0, dist.max() + dist_bin, dist_bin)
        _no_edge_hits(dist, d_edges)
        conf.get_distance_from_latlon = lambda _ds: dist
        nasc = conf._get_expected_nasc_val_nanmean(ds, dist_bin, nasc_range_bin, C)
        # depth = echo_range + offset taken BEFORE NaNs were put into echo_range: stored as the offset and the mask
        off = np.nanmax(depth - np.where(np.isnan(g[f"{tag}_echo_range"]), np.nan, g[f"{tag}_echo_range"]))
        g[f"{tag}_depth_offset"], g[f"{tag}_distance_nmi"] = np.array(off), dist
        g[f"{tag}_depth_at_nan_range"] = depth[np.isnan(g[f"{tag}_echo_range"])]
        g[f"{tag}_dist_bin"], g[f"{tag}_nasc_range_bin"] = np.array(float(dist_bin)), np.array(float(nasc_range_bin))
        g[f"{tag}_nasc"] = np.asarray(nasc.data if hasattr(nasc, "data") else nasc)
        print(tag, "MVBS", g[f"{tag}_mvbs"].shape, "NaN", int(np.isnan(g[f"{tag}_mvbs"]).sum()), "NASC",
              g[f"{tag}_nasc"].shape, "NaN", int(np.isnan(g[f"{tag}_nasc"]).sum()))

    # ---- the reference's own small fixtures (conftest.py:121-165): regular, and irregular + jitter + NaNs
    rng = np.random.default_rng(2026)
    ds = mock._gen_Sv_echo_range_regular(**params, ping_time_jitter_max_ms=0)
    ds._vars["Sv"].data = sample.copy()
    depth = ds["echo_range"].data + 2.5  # add_depth(depth_offset=2.5), consolidate/api.py:226
    add("small_regular", ds, "1s", 2, rng.uniform(0.05, 0.2, params["ping_time_len"]), 0.5, 2, depth)

    np.random.seed(30)  # the generator draws its jitter from the global NumPy state (mock_data.py:22)
    ds = mock._gen_Sv_echo_range_irregular(**params, depth_interval=[0.5, 0.32, 0.2], depth_ping_time_len=[2, 3, 5],
                                           ping_time_jitter_max_ms=30)
    ds._vars["Sv"].data = sample.copy()
    depth = ds["echo_range"].data + 2.5  # depth is added BEFORE the NaNs are sprinkled (conftest.py:152-165)
    for pos in nan_ilocs:
        ds._vars["echo_range"].data[pos] = np.nan
        ds._vars["Sv"].data[pos] = np.nan
    add("small_irregular", ds, "1s", 2, rng.uniform(0.05, 0.2, params["ping_time_len"]), 0.5, 2, depth)

    # ---- larger seeded datasets from the same generators
    ds = mock._gen_Sv_echo_range_regular(channel_len=2, depth_len=60, depth_interval=0.5, ping_time_len=200,
                                         ping_time_interval="0.37s", random_number_generator=np.random.default_rng(11))
    ds._vars["Sv"].data = -90 + 40 * ds["Sv"].data  # dB-like spread (the generator draws U(0, 1))
    add("regular", ds, "5s", 2, rng.uniform(0.004, 0.02, 200), 0.5, 2, ds["echo_range"].data + 7.0)

    for tag, value_nans in (("irregular", 0), ("irregular_valnan", 9)):
        np.random.seed(50)
        ds = mock._gen_Sv_echo_range_irregular(channel_len=2, depth_len=60, ping_time_len=240,
                                               depth_ping_time_len=[40, 120, 80], ping_time_jitter_max_ms=50,
                                               random_number_generator=np.random.default_rng(12))
        ds._vars["Sv"].data = -90 + 40 * ds["Sv"].data
        depth = ds["echo_range"].data + 1.0
        r2 = np.random.default_rng(13)
        holes = r2.random(ds["Sv"].shape) < 0.03
        ds._vars["echo_range"].data[holes] = np.nan  # NaN coordinates drop the sample from every bin
        ds._vars["Sv"].data[holes] = np.nan
        # NaN VALUES under a valid coordinate: _get_expected_mvbs_val averages with np.mean, so a bin holding one is
        # NaN -- the skipna=False answer of compute_MVBS; the NASC expectation uses nanmean (skipna=True)
        for _ in range(value_nans):
            ds._vars["Sv"].data[r2.integers(2), r2.integers(240), r2.integers(60)] = np.nan
        steps = np.random.default_rng(14).uniform(0.002, 0.012, 240)
        add(tag, ds, "10s", 2, steps, 0.5, 2, depth)
    # the second variant differs from the first only by its value NaNs: keep their positions, not the arrays again
    g["irregular_valnan_positions"] = np.argwhere(np.isnan(g["irregular_valnan_Sv"]) & ~np.isnan(g["irregular_Sv"]))
    for k in ("Sv", "echo_range", "ping_time", "distance_nmi", "depth_at_nan_range", "mvbs_time_labels"):
        assert k == "Sv" or np.array_equal(g[f"irregular_valnan_{k}"], g[f"irregular_{k}"], equal_nan=True)
        del g[f"irregular_valnan_{k}"]

    # ---- the Echoview known answer (test_commongrid_api.py:155-167): dataset by the reference's _create_dataset
    class _Rng:  # a generator whose draw is recorded (conftest.py:412 adds rng.random() * 5 to the linear sv)
        def __init__(self):
            self.r = np.random.default_rng(99)

        def random(self):
            return self.r.random()

    dim0 = np.array([0.5, 1.5, 2.5, 3.5, 9])
    sv0 = np.array([[1.0, 2.0, 3.0, 4.0, np.nan], [6.0, 7.0, 8.0, 9.0, 10.0], [11.0, 12.0, 13.0, 14.0, 15.0],
                    [16.0, 17.0, 18.0, 19.0, np.nan], [21.0, 22.0, 23.0, 24.0, 25.0]])
    shared = _Rng()
    chans = [conf._create_dataset(i, sv0, dim0, shared) for i in range(2)]
    Sv = np.stack([c["Sv"].data for c in chans])            # (channel, range_sample, distance_nmi)
    dep = np.stack([c["depth"].data for c in chans])
    both = xr_shim.Dataset(coords={"channel": np.array(["ch_0", "ch_1"]), "range_sample": np.arange(5),
                                   "distance_nmi": np.arange(5)})
    both["Sv"] = xr_shim.DataArray(Sv, dims=["channel", "range_sample", "distance_nmi"])
    both["depth"] = xr_shim.DataArray(dep, dims=["channel", "range_sample", "distance_nmi"])
    g["echoview_Sv"], g["echoview_depth"] = Sv, dep
    g["echoview_nasc"] = np.array([conf.get_NASC_echoview(both, ch_idx=i, r0=2, r1=20) for i in range(2)])
    print("echoview", g["echoview_nasc"])
    np.savez_compressed(OUT, **g)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
