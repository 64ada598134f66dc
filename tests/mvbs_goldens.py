"""Loader of tests/golden/ref_mvbs_goldens.npz: the reference's own brute-force expectations of compute_MVBS and
compute_NASC (mock_data.py:28-85, tests/commongrid/conftest.py:466-617), executed from /root/reference by
oracle/gen_mvbs_goldens.py on datasets made by the reference's generators.  Data only."""
import os

import numpy as np

PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_mvbs_goldens.npz")
CASES = ["small_regular", "small_irregular", "regular", "irregular", "irregular_valnan"]


def load(tag):
    """dict(Sv, echo_range, depth, ping_time, distance_nmi, ping_time_bin, range_bin, dist_bin, nasc_range_bin,
    mvbs, mvbs_time_labels, nasc, value_nans) of one case."""
    g = np.load(PATH)
    base = "irregular" if tag == "irregular_valnan" else tag
    d = {k: g[f"{base}_{k}"] for k in ("Sv", "echo_range", "ping_time", "distance_nmi", "depth_at_nan_range",
                                       "mvbs_time_labels")}
    d["Sv"], d["echo_range"] = d["Sv"].copy(), d["echo_range"].copy()
    d["value_nans"] = 0
    if tag == "irregular_valnan":  # the same dataset with a few NaN VALUES under valid coordinates
        pos = g["irregular_valnan_positions"]
        d["Sv"][tuple(pos.T)] = np.nan
        d["value_nans"] = len(pos)
    for k in ("ping_time_bin", "range_bin", "dist_bin", "nasc_range_bin", "depth_offset", "mvbs", "nasc"):
        d[k] = g[f"{tag}_{k}"]
    d["ping_time_bin"] = str(d["ping_time_bin"])
    for k in ("range_bin", "dist_bin", "nasc_range_bin", "depth_offset"):
        d[k] = float(d[k])
    # depth = echo_range + offset, taken before NaNs were put into echo_range
    nanpos = np.isnan(d["echo_range"])
    d["depth"] = d["echo_range"] + d["depth_offset"]
    d["depth"][nanpos] = d.pop("depth_at_nan_range")
    return d


def compare_mvbs(got, exp, tol=1e-10):
    """The reference's expectation lays its range edges as arange(0, max + 2, bin), compute_MVBS as
    arange(0, max + bin, bin) (mock_data.py:58 vs commongrid/api.py:115): equal for the 2 m bins of the fixtures;
    compared over the common bins otherwise, the surplus ones must be empty."""
    assert got.shape[:2] == exp.shape[:2], (got.shape, exp.shape)
    n = min(got.shape[2], exp.shape[2])
    assert np.isnan(got[..., n:]).all() and np.isnan(exp[..., n:]).all()
    g, e = got[..., :n], exp[..., :n]
    np.testing.assert_array_equal(np.isnan(g), np.isnan(e))
    np.testing.assert_allclose(g, e, rtol=tol, atol=tol, equal_nan=True)  # test_commongrid_api.py:432
