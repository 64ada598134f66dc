"""Oracle for the noise masks (oracle/masks.py) against (a) the outputs of the reference's own leaf
functions (tests/golden/ref_mask_goldens.npz, made by oracle/gen_mask_goldens.py) and (b) the
reference's structural tests restated on synthetic scenes (tests/clean/test_noise.py)."""
import os

import numpy as np
import pytest

from oracle import masks as omask

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_mask_goldens.npz")


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLDEN)


def _scene(C, P, S, seed, step=0.3):
    rng = np.random.default_rng(seed)
    sv = -70 + 4 * rng.standard_normal((C, P, S))
    sv[rng.random((C, P, S)) < 0.03] += 30
    sv[rng.random((C, P, S)) < 0.04] = np.nan
    depth = 2.0 + np.arange(S)[None, None, :] * (step * (1 + 0.4 * np.arange(C)))[:, None, None] + np.zeros((C, P, 1))
    return sv, depth


def test_echopy_leaf_functions_match_reference_outputs(gold):
    for i in range(4):
        n, thr = gold[f"imp{i}_args"]
        np.testing.assert_array_equal(
            omask.echopy_impulse_noise_mask(gold[f"imp{i}_sv"], int(n), thr), gold[f"imp{i}_mask"])
        up, lw, n, thr = gold[f"att{i}_args"]
        np.testing.assert_array_equal(
            omask.echopy_attenuated_signal_mask(gold[f"att{i}_sv"], gold[f"att{i}_range"], up, lw, int(n), thr),
            gold[f"att{i}_mask"])
    np.testing.assert_array_equal(omask._lin(gold["db_in"]), gold["log2lin"])
    np.testing.assert_array_equal(omask._log(gold["log2lin"]), gold["lin2log"])


def test_index_binning_pool_matches_symmetric_pad_windows():
    """tests/clean/test_noise.py:342-441: generic_filter(reflect) == aggregate over np.pad(symmetric)."""
    sv, depth = _scene(2, 20, 30, 1)
    n, exclude_above = 2, 4.0
    for f in (np.nanmean, np.nanmedian):
        pooled = omask.index_binning_pool_Sv(sv, depth, f, 1.0, n, exclude_above)
        m = omask.nsamples_per_bin(depth, 1.0)
        s0 = int(np.argmin(depth <= exclude_above))
        assert 0 < s0 < 30 and np.isnan(pooled[:, :, :s0]).all()
        for c in range(2):
            pad = np.pad(sv[c, :, s0:], ((n, n), (m[c], m[c])), mode="symmetric")
            for p in range(20):
                for s in range(30 - s0):
                    w = 10 ** (pad[p:p + 2 * n + 1, s:s + 2 * m[c] + 1] / 10)
                    np.testing.assert_allclose(pooled[c, p, s0 + s], 10 * np.log10(f(w)), rtol=1e-10, atol=1e-10)


def test_index_binning_downsample_upsample_blocks():
    """tests/clean/test_noise.py:616-683: every sample carries the linear mean of its n_c-sample block."""
    sv, depth = _scene(3, 7, 53, 2)
    up = omask.index_binning_downsample_upsample(sv, depth, 2.0)
    n = omask.nsamples_per_bin(depth, 2.0)
    assert list(n) == [int(np.ceil(2.0 / (0.3 * (1 + 0.4 * c)))) for c in range(3)]
    for c in range(3):
        for b in range(0, 53, n[c]):
            blk = 10 ** (sv[c, :, b:b + n[c]] / 10)
            with np.errstate(invalid="ignore", divide="ignore"):
                exp = 10 * np.log10(np.nansum(blk, axis=1) / np.sum(~np.isnan(blk), axis=1))
            for s in range(b, min(b + n[c], 53)):
                np.testing.assert_allclose(up[c, :, s], exp, rtol=1e-12)


def test_downsample_upsample_bins_are_left_closed():
    """tests/clean/test_noise.py:550-605: flox bins [left, right) and the up-sampled value of a
    sample is the down-sampled value of the bin holding its depth."""
    sv, depth = _scene(2, 5, 40, 3)
    down, up = omask.downsample_upsample(sv, depth, 2.5)
    edges = np.arange(np.nanmin(depth), np.nanmax(depth) + 2.5, 2.5)
    for c in range(2):
        for s in range(40):
            j = np.searchsorted(edges, depth[c, 0, s], side="right") - 1
            assert edges[j] <= depth[c, 0, s] < edges[j + 1]
            np.testing.assert_array_equal(up[c, :, s], down[c, :, j])


def test_pool_Sv_boundaries_are_nan():
    """tests/clean/test_noise.py:161-219: windows that leave the ping / depth domain are not pooled."""
    sv, depth = _scene(1, 12, 25, 4)
    pooled = omask.pool_Sv(sv, depth, np.nanmean, 1.0, 2, 3.0)
    assert np.isnan(pooled[:, :2]).all() and np.isnan(pooled[:, 11:]).all()
    assert np.isfinite(pooled[:, 2:11]).any()
    d = depth[0, 0]
    inside = (d - 1.0 >= max(d.min(), 3.0)) & (d + 1.0 <= d.max())
    assert np.isnan(pooled[0, 5, ~inside]).all() and np.isfinite(pooled[0, 5, inside]).all()
    p, s = 5, int(np.flatnonzero(inside)[3])
    w = (np.abs(depth[0, 3:8] - d[s]) <= 1.0)
    exp = 10 * np.log10(np.nanmean(np.where(w, 10 ** (sv[0, 3:8] / 10), np.nan)))
    np.testing.assert_allclose(pooled[0, p, s], exp, rtol=1e-12)


def test_mask_wrappers_shapes_errors_and_apply_mask():
    sv, depth = _scene(2, 12, 30, 5)
    imp = omask.mask_impulse_noise(sv, depth, "2m", 2, "10.0dB", True)
    assert imp.shape == (2, 30, 12) and imp.dtype == bool  # (channel, range_sample, ping_time)
    tr = omask.mask_transient_noise(sv, depth, depth_bin="1m", num_side_pings=2, exclude_above="3.0m",
                                    transient_noise_threshold="8.0dB", use_index_binning=True)
    assert tr.shape == sv.shape
    with pytest.raises(ValueError, match="must be `nanmean` or `nanmedian`"):
        omask.mask_transient_noise(sv, depth, func="nanmode")
    assert not omask.mask_attenuated_signal(sv, depth, "1800.0m", "2800.0m").any()
    with pytest.raises(ValueError, match="Minimum range"):
        omask.mask_attenuated_signal(sv, depth, "180.0m", "170.0m")
    out = omask.apply_mask(sv, [~tr, np.ones((12, 30))], fill_value=-1.0)
    np.testing.assert_array_equal(out, np.where(~tr, sv, -1.0))


API_GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_maskapi_goldens.npz")


def _kw(gold, tag):
    import ast

    return {k: ast.literal_eval(v) for k, v in (s.split("=", 1) for s in gold[f"{tag}_kw"].tolist())}


@pytest.mark.parametrize("tag", ["tri0", "tri1", "tri2", "trv0", "trv1", "imp0", "imp1", "imp2", "att0", "att1", "att2"])
def test_mask_api_matches_reference_end_to_end(tag):
    """The reference's own clean/api.py mask functions, executed end to end (oracle/gen_maskapi_goldens.py):
    string parsing, samples-per-bin, pooling / up-sampling, start index, early return, dimension order."""
    gold = np.load(API_GOLDEN)
    sv, er, kw = gold[f"{tag}_Sv"], gold[f"{tag}_echo_range"], _kw(gold, tag)
    if tag.startswith("tr"):
        got = omask.mask_transient_noise(sv, er, use_index_binning=tag.startswith("tri"), **kw)
        assert gold[f"{tag}_dims"].tolist() == ["channel", "ping_time", "range_sample"]
    elif tag.startswith("imp"):
        got = omask.mask_impulse_noise(sv, er, use_index_binning=True, **kw)
        assert gold[f"{tag}_dims"].tolist() == ["channel", "range_sample", "ping_time"]
    else:
        got = omask.mask_attenuated_signal(sv, er, **kw)
        assert gold[f"{tag}_dims"].tolist() == ["channel", "ping_time", "range_sample"]
    np.testing.assert_array_equal(got, gold[f"{tag}_mask"])
