"""Kernel-level parity of the noise-mask kernels through the C ABI vs the oracle
and vs the outputs of the reference's own leaf functions (tests/golden/ref_mask_goldens.npz).

Masks are boolean: they must be IDENTICAL wherever the oracle's decision margin
|difference - threshold| exceeds the floating-point noise of the compared quantity (1e-9 dB in
fp64, 1e-3 dB in fp32); the pooled / smoothed Sv they derive from are held to the usual tolerances.
"""
import os

import numpy as np
import pytest

from oracle import masks as omask

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_mask_goldens.npz")
RTOL = {"float64": 1e-9, "float32": 1e-3}
MARGIN = {"float64": 1e-9, "float32": 2e-3}


@pytest.fixture(scope="module")
def env():
    import torch

    if not torch.cuda.is_available():
        pytest.fail("these tests need a GPU (run with -m 'not gpu' on CPU boxes)")
    from echopype_amd import ops

    return torch, ops


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLDEN)


def _dev(torch, a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).cuda()
    return t.to(dtype) if dtype is not None else t


def _close(got, exp, rtol, what=""):
    got, exp = np.asarray(got, dtype=np.float64), np.asarray(exp, dtype=np.float64)
    assert got.shape == exp.shape
    np.testing.assert_array_equal(np.isnan(got), np.isnan(exp), err_msg=f"{what}: NaN pattern")
    fin = np.isfinite(exp)
    np.testing.assert_array_equal(got[~fin & ~np.isnan(exp)], exp[~fin & ~np.isnan(exp)])
    err = np.abs(got[fin] - exp[fin]) / np.maximum(np.abs(exp[fin]), 1.0)
    assert err.size == 0 or err.max() <= rtol, f"{what}: max rel err {err.max():.3e} > {rtol}"


def _scene(C, P, S, seed, step=0.19, nan_frac=0.03, spikes=True, ragged=False):
    """Sv (dB) with a gradient, impulses, transient blobs, attenuated pings, NaNs; depth per channel."""
    rng = np.random.default_rng(seed)
    sv = -70 + 4 * rng.standard_normal((C, P, S)) - 10 * np.linspace(0, 1, S)[None, None, :]
    if spikes:
        sv[rng.random((C, P, S)) < 0.02] += 30
        for _ in range(3):
            c, p, s = rng.integers(C), rng.integers(P), rng.integers(S)
            sv[c, max(p - 1, 0):p + 2, max(s - 8, 0):s + 8] += 25
        sv[:, rng.random(P) < 0.1, :] -= 15
    sv[rng.random((C, P, S)) < nan_frac] = np.nan
    steps = step * (1 + 0.37 * np.arange(C))
    depth = 1.5 + np.arange(S)[None, None, :] * steps[:, None, None] + 0.3 * rng.random((C, P, 1))
    if ragged:
        sv[-1, :, S - S // 5:] = np.nan
        depth[-1, :, S - S // 5:] = np.nan
    return sv, depth


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_range_bin_smooth_index_mode(env, dtype):
    torch, ops = env
    sv, depth = _scene(3, 17, 203, 1, ragged=True)
    sv, depth = sv.astype(dtype), depth.astype(dtype)
    exp = omask.index_binning_downsample_upsample(sv.astype(np.float64), depth.astype(np.float64), 2.0)
    n = omask.nsamples_per_bin(depth.astype(np.float64), 2.0)
    assert len(set(n.tolist())) == 3  # channel-specific block lengths
    svt = _dev(torch, sv)
    for c in range(3):
        got = ops.range_bin_smooth(svt[c:c + 1].contiguous(), nper=int(n[c])).cpu().numpy()
        _close(got[0], exp[c], RTOL[dtype], f"channel {c}")


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_range_bin_smooth_value_mode(env, dtype):
    torch, ops = env
    sv, depth = _scene(2, 13, 150, 2)
    sv[0, 3, :] = np.nan  # an all-NaN ping -> every bin NaN
    sv, depth = sv.astype(dtype), depth.astype(dtype)
    d64 = depth.astype(np.float64)
    down, exp = omask.downsample_upsample(sv.astype(np.float64), d64, 5.0)
    r0 = float(np.nanmin(d64))
    nb = len(np.arange(r0, np.nanmax(d64) + 5.0, 5.0)) - 1
    assert down.shape[-1] == nb
    got = ops.range_bin_smooth(_dev(torch, sv), range=_dev(torch, depth), r0=r0, bin=5.0, nbins=nb)
    _close(got.cpu().numpy(), exp, RTOL[dtype])


def test_range_bin_smooth_value_mode_edges_and_nan_depth(env):
    """Samples exactly on bin edges go to the bin they open ([e_j, e_j+1)); the global maximum falls
    outside the last left-closed interval when it is an exact multiple; NaN depth -> last bin."""
    torch, ops = env
    S = 41
    depth = np.tile(10.0 + 0.5 * np.arange(S), (1, 2, 1))  # edges 10, 12.5, 15 ... hit exactly
    depth[0, 1, -6:] = np.nan
    rng = np.random.default_rng(0)
    sv = -60 + 5 * rng.standard_normal((1, 2, S))
    down, exp = omask.downsample_upsample(sv, depth, 2.5)
    r0, nb = 10.0, len(np.arange(10.0, np.nanmax(depth) + 2.5, 2.5)) - 1
    got = ops.range_bin_smooth(_dev(torch, sv), range=_dev(torch, depth), r0=r0, bin=2.5, nbins=nb)
    _close(got.cpu().numpy(), exp, 1e-9)


@pytest.mark.parametrize("mode", ["value", "index"])
def test_range_bin_smooth_more_bins_than_the_lds_holds(env, mode):
    """A ping with ~14 000+ bins (bins finer than the samples): the kernel takes the bins in segments -- same result
    as the oracle (every bin holds one sample or none: smoothing is the identity where defined)."""
    torch, ops = env
    rng = np.random.default_rng(3)
    if mode == "value":
        S = 3000
        depth = np.tile(5.0 + 0.05 * np.arange(S), (1, 3, 1))
        sv = -60 + 5 * rng.standard_normal((1, 3, S))
        sv[0, 1, 100:140] = np.nan
        down, exp = omask.downsample_upsample(sv, depth, 0.01)
        r0, nb = 5.0, len(np.arange(5.0, np.nanmax(depth) + 0.01, 0.01)) - 1
        assert nb > 13500
        got = ops.range_bin_smooth(_dev(torch, sv), range=_dev(torch, depth), r0=r0, bin=0.01, nbins=nb)
    else:
        S = 30_001
        sv = -60 + 5 * rng.standard_normal((1, 2, S))
        sv[0, 0, 7::11] = np.nan
        depth = np.tile(np.arange(S) * 1.0, (1, 2, 1))
        exp = omask.index_binning_downsample_upsample(sv, depth, 2.0)  # 2 samples per bin -> 15 001 bins
        got = ops.range_bin_smooth(_dev(torch, sv), nper=2)
    _close(got.cpu().numpy(), exp, 1e-9)


def test_impulse_mask_reference_goldens(env, gold):
    torch, ops = env
    for i in range(4):
        sv = gold[f"imp{i}_sv"]  # (range_sample, ping_time)
        n, thr = gold[f"imp{i}_args"]
        up = _dev(torch, sv.T[None])  # (1, P, S)
        got = ops.impulse_mask(up, int(n), float(thr)).cpu().numpy()[0].T.astype(bool)
        np.testing.assert_array_equal(got, gold[f"imp{i}_mask"], err_msg=f"case {i}")


def test_impulse_mask_more_side_pings_than_pings(env):
    torch, ops = env
    up = _dev(torch, np.zeros((1, 3, 5)))
    assert ops.impulse_mask(up, 7, 10.0).cpu().numpy().all()  # both sides missing -> inf > thr


def test_attenuated_mask_reference_goldens(env, gold):
    torch, ops = env
    for i in range(4):
        sv, rg = gold[f"att{i}_sv"], gold[f"att{i}_range"]
        up, lw, n, thr = gold[f"att{i}_args"]
        got = ops.attenuated_mask(_dev(torch, sv[None]), _dev(torch, rg[None]), up, lw, int(n), thr)
        np.testing.assert_array_equal(got.cpu().numpy()[0].astype(bool), gold[f"att{i}_mask"],
                                      err_msg=f"case {i}")


@pytest.mark.parametrize("n", [1, 6, 15])
def test_attenuated_mask_sliding_blocks_vs_oracle(env, n):
    """The block median carried from ping to ping (attenuated_slide_kernel): several 128-ping chunks, layer limits
    that change along the pings (a platform that sinks by one range step every 90 pings), runs of NaN pings, runs of
    equal values (more candidates in the median's bin than its LDS list holds -> the sweeping path), -inf samples,
    attenuated pings on both sides of the threshold."""
    torch, ops = env
    C, P, S = 2, 420, 260
    rng = np.random.default_rng(100 + n)
    sv, depth = _scene(C, P, S, 100 + n, step=0.5, spikes=False)
    depth = depth[:, :1, :] + 0.5 * (np.arange(P) // 90)[None, :, None]      # identical rows that shift in steps
    sv[:, rng.random(P) < 0.15, :] -= rng.uniform(2, 9)                      # attenuated pings
    sv[0, 100:104, :] = np.nan
    sv[1, 200:260, 40:200] = -63.25                                          # ~10 000 equal values in a block
    sv[0, 300, 50:90] = -np.inf
    got = ops.attenuated_mask(_dev(torch, sv), _dev(torch, depth), 30.0, 110.0, n, -4.0).cpu().numpy().astype(bool)
    exp = np.stack([omask.echopy_attenuated_signal_mask(sv[c], depth[c], 30.0, 110.0, n, -4.0) for c in range(C)])
    np.testing.assert_array_equal(got, exp)
    assert exp.any() and not exp.all()


def test_attenuated_mask_layers_beyond_the_lds_ring(env):
    """Layers longer than the 1024 samples a lane prefetches, and blocks of more than 20 480 values: the sliding kernel
    takes the sweeping selection for those pings (same kernel, other branch)."""
    torch, ops = env
    C, P, S = 1, 70, 1500
    sv, depth = _scene(C, P, S, 321, step=0.1, spikes=False)
    rng = np.random.default_rng(9)
    sv[:, rng.random(P) < 0.2, :] -= 7.0
    for n, lo, hi in ((6, 10.0, 130.0), (12, 20.0, 110.0)):   # 1200-sample layer; 24 x 900 = 21 600 values per block
        got = ops.attenuated_mask(_dev(torch, sv), _dev(torch, depth), lo, hi, n, -4.0).cpu().numpy().astype(bool)
        exp = np.stack([omask.echopy_attenuated_signal_mask(sv[c], depth[c], lo, hi, n, -4.0) for c in range(C)])
        np.testing.assert_array_equal(got, exp)
        assert exp.any() and not exp.all()


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_attenuated_mask_vs_oracle(env, dtype):
    torch, ops = env
    sv, depth = _scene(3, 80, 300, 5, step=0.5, ragged=True)
    sv[1, 40:44, :] = np.nan
    sv[0, 10, 20:60] = -np.inf
    sv, depth = sv.astype(dtype), depth.astype(dtype)
    got = ops.attenuated_mask(_dev(torch, sv), _dev(torch, depth), 30.0, 90.0, 6, -5.0).cpu().numpy()
    exp = np.stack([omask.echopy_attenuated_signal_mask(sv[c].astype(np.float64), depth[c],
                                                        np.dtype(dtype).type(30.0), np.dtype(dtype).type(90.0),
                                                        6, -5.0) for c in range(3)])
    diff = got.astype(bool) != exp
    if dtype == "float64":
        assert not diff.any()
    else:  # a ping whose median difference sits within fp32 noise of the threshold may flip
        assert diff.any(axis=2).sum() <= 1
    assert exp.any() and not exp.all()


@pytest.mark.parametrize("dtype", ["float64", "float32"])
@pytest.mark.parametrize("func", ["nanmean", "nanmedian"])
def test_pool_sv_vs_generic_filter(env, dtype, func):
    """The reference's own structural test (tests/clean/test_noise.py:342-441) with
    scipy.ndimage.generic_filter(mode="reflect") as the engine."""
    torch, ops = env
    C, P, S = 2, 23, 70
    sv, depth = _scene(C, P, S, 7, step=0.4)
    sv[0, 5:9, 30:50] = np.nan  # a window with no valid sample at all
    sv, depth = sv.astype(dtype), depth.astype(dtype)
    f = np.nanmean if func == "nanmean" else np.nanmedian
    d64 = depth.astype(np.float64)
    m = omask.nsamples_per_bin(d64, 2.0)
    exclude_above = 6.0
    s0 = int(np.argmin(d64 <= exclude_above))
    assert 0 < s0 < S
    exp = omask.index_binning_pool_Sv(sv.astype(np.float64), d64, f, 2.0, 3, exclude_above)
    thr = 8.0
    svt = _dev(torch, sv)
    for c in range(C):
        pooled, mask = ops.pool_sv(svt[c:c + 1].contiguous(), s0, 3, int(m[c]), func=func, threshold=thr)
        _close(pooled.cpu().numpy()[0], exp[c], RTOL[dtype], f"{func} channel {c}")
        with np.errstate(invalid="ignore"):
            margin = sv[c].astype(np.float64) - exp[c] - thr
        sure = ~(np.abs(margin) < MARGIN[dtype])
        np.testing.assert_array_equal(mask.cpu().numpy()[0].astype(bool)[sure], (margin > 0)[sure])
        assert (margin > 0).any()


def test_pool_sv_window_larger_than_array(env):
    """reflect is periodic with period 2N: windows wider than the data wrap several times."""
    torch, ops = env
    sv, _ = _scene(1, 5, 9, 11, spikes=False, nan_frac=0.1)
    import scipy.ndimage

    for func, f in (("nanmean", np.nanmean), ("nanmedian", np.nanmedian)):
        exp = 10 * np.log10(scipy.ndimage.generic_filter(10 ** (sv[0] / 10), f, size=[2 * 7 + 1, 2 * 11 + 1],
                                                         mode="reflect"))
        pooled, _ = ops.pool_sv(_dev(torch, sv), 0, 7, 11, func=func)
        _close(pooled.cpu().numpy()[0], exp, 1e-9, func)


def test_pool_sv_sliding_sum_keeps_small_values_after_a_huge_one(env):
    """The ping pass carries the window sum from ping to ping (enter / leave); a value 10^14 times larger than its
    neighbours passing through the window must leave nothing behind (double-double carry), a +inf must not turn
    the pings after it into NaN, and segment joins (512 pings) must be seamless."""
    import scipy.ndimage

    torch, ops = env
    rng = np.random.default_rng(5)
    P, S, n, m = 1300, 24, 6, 4
    sv = -120 + 3 * rng.standard_normal((1, P, S))
    sv[0, 40, 5] = 25.0          # 10^14.5 above the background
    sv[0, 700:703, 10:14] = 40.0
    sv[0, 511, 3] = sv[0, 512, 17] = 30.0   # at a segment join
    sv[0, rng.random((P, S)) < 0.05] = np.nan
    with np.errstate(invalid="ignore"):
        exp = 10 * np.log10(scipy.ndimage.generic_filter(10 ** (sv[0] / 10), np.nanmean, size=[2 * n + 1, 2 * m + 1],
                                                         mode="reflect"))
    pooled, _ = ops.pool_sv(_dev(torch, sv), 0, n, m)
    got = pooled.cpu().numpy()[0]
    _close(got, exp, 1e-9, "sliding window after a spike")
    assert np.nanmin(exp) < -115 and np.nanmax(exp) > 0  # both regimes present
    sv[0, 100, 7] = np.inf
    pooled, _ = ops.pool_sv(_dev(torch, sv), 0, n, m)
    got2 = pooled.cpu().numpy()[0]
    hit = np.zeros((P, S), bool)
    hit[100 - n:100 + n + 1, 7 - m:7 + m + 1] = True
    assert np.isposinf(got2[hit]).all()
    np.testing.assert_array_equal(got2[~hit], got[~hit])


@pytest.mark.parametrize("dtype", ["float64", "float32"])
@pytest.mark.parametrize("depth_kind", ["one vector", "limits change", "heave"])
def test_attenuated_mask_carried_block_equals_medians_from_memory(env, dtype, depth_kind):
    """The block carried from ping to ping (attenuated_prepare_kernel + attenuated_walk_kernel) against the kernel that
    takes both medians of every ping from memory (reached with S % 4 != 0: one padding sample far below the layer):
    attenuated pings, NaN, an all-NaN stretch, a flat stretch (more than 64 values in the median's bin and inside one
    16-bit code of a ping's own layer), layer limits that change once / from ping to ping / move by a sample or two at
    every ping (a depth with heave: the layer is shifted inside the ring), several chunks of pings."""
    from echopype_amd import _lib

    torch, ops = env
    rng = np.random.default_rng(31)
    C, P, S, n = 2, 1300, 400, 7
    sv = -70 + 4 * rng.standard_normal((C, P, S)) - 10 * np.linspace(0, 1, S)
    att = rng.random((C, P)) < 0.05
    sv[att] -= rng.uniform(3, 30, size=att.sum())[:, None]
    sv[rng.random((C, P, S)) < 0.03] = np.nan
    sv[0, 500:520] = np.nan
    sv[1, 900:960, 60:300] = -71.25
    depth = np.broadcast_to(1.0 + 0.5 * np.arange(S), (C, P, S)).copy()
    if depth_kind == "limits change":
        depth[1, 700:] *= 1.1
        depth[0, 1000:1010] *= 1 + 0.05 * rng.random((10, 1))
    elif depth_kind == "heave":
        depth = depth + 3.0 * np.sin(np.arange(P) / 7.0)[None, :, None]
    svt, rgt = _dev(torch, sv.astype(dtype)), _dev(torch, depth.astype(dtype))
    with _lib.launch_trace() as tr:
        new = ops.attenuated_mask(svt, rgt, 40.0, 140.0, n, -6.0)
    assert "attenuated_walk_kernel" in tr.kernels and "attenuated_prepare_kernel" in tr.kernels
    pad = torch.nn.functional.pad
    with _lib.launch_trace() as tr:
        ref = ops.attenuated_mask(pad(svt, (0, 1), value=float("nan")).contiguous(),
                                  pad(rgt, (0, 1), value=1.0e6).contiguous(), 40.0, 140.0, n, -6.0)[:, :, :S]
    assert tr.kernels == ["attenuated_mask_kernel"]
    a, b = new.cpu().numpy(), ref.cpu().numpy()
    np.testing.assert_array_equal(a, b)
    assert 20 < a[:, :, 0].sum() < 0.2 * C * P and (a == a[:, :, :1]).all()


def _median_filter_db(sv2d, n, m):
    import scipy.ndimage

    with np.errstate(invalid="ignore", divide="ignore", over="ignore"):
        lin = 10 ** (sv2d.astype(np.float64) / 10)
        return 10 * np.log10(scipy.ndimage.generic_filter(lin, np.nanmedian, size=[2 * n + 1, 2 * m + 1],
                                                          mode="reflect"))


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_pool_sv_median_window_carried_from_ping_to_ping(env, dtype):
    """nanmedian pooling walks the pings with the window in an LDS ring and a histogram kept up to date (entering /
    leaving row): three 512-ping segments, values below and above the histogram's [-256, 0) dB span (clamped bins),
    +-inf, 8 % NaN, a block with no valid value, even and odd counts -- against scipy's generic_filter + np.nanmedian,
    the engine the reference's own test uses (tests/clean/test_noise.py:342-441)."""
    from echopype_amd import _lib

    torch, ops = env
    rng = np.random.default_rng(21)
    P, S, n, m = 1100, 24, 6, 4
    sv = -80 + 3 * rng.standard_normal((1, P, S))
    sv[0, rng.random((P, S)) < 0.04] = -300 - 20 * rng.random()
    sv[0, rng.random((P, S)) < 0.04] = 5 + 20 * rng.random()
    sv[0, 200:260, 3:9] = -330 + 30 * rng.random((60, 6))        # windows whose median lies in the lowest bin
    sv[0, 600:650, 12:20] = 10 + 30 * rng.random((50, 8))        # ... in the highest bin
    sv[0, 505:520, 5] = np.inf
    sv[0, 900, 10:14] = -np.inf
    sv[0, rng.random((P, S)) < 0.08] = np.nan
    sv[0, 300:320, :] = np.nan
    sv = sv.astype(dtype)
    thr = 6.0
    with _lib.launch_trace() as tr:
        pooled, mask = ops.pool_sv(_dev(torch, sv), 2, n, m, func="nanmedian", threshold=thr)
    assert "pool_median_slide_kernel" in tr.kernels
    got = pooled.cpu().numpy()[0]
    assert np.isnan(got[:, :2]).all() and not mask.cpu().numpy()[0, :, :2].any()
    exp2 = _median_filter_db(sv[0, :, 2:], n, m)  # pooled from the first sample on: the reflect domain starts there
    _close(got[:, 2:], exp2, RTOL[dtype], "carried median")
    assert np.isnan(exp2).any() and (exp2[np.isfinite(exp2)] < -256).any() and (exp2[np.isfinite(exp2)] > 0).any()
    with np.errstate(invalid="ignore"):
        margin = sv[0, :, 2:].astype(np.float64) - exp2 - thr
    sure = ~(np.abs(margin) < MARGIN[dtype])
    np.testing.assert_array_equal(mask.cpu().numpy()[0, :, 2:].astype(bool)[sure], (margin > 0)[sure])


def test_pool_sv_median_of_a_flat_field_uses_the_radix_selection(env):
    """625 values within 0.01 dB: the median's histogram bin holds the whole window (more than the 512 candidates the
    counting rank takes), every step falls back to the radix selection over the ring -- same answer."""
    torch, ops = env
    rng = np.random.default_rng(22)
    P, S, n, m = 30, 30, 12, 12
    sv = -70.003 + 0.01 * rng.random((1, P, S))
    sv[0, rng.random((P, S)) < 0.05] = np.nan
    exp = _median_filter_db(sv[0], n, m)
    pooled, _ = ops.pool_sv(_dev(torch, sv), 0, n, m, func="nanmedian")
    _close(pooled.cpu().numpy()[0], exp, 1e-12, "flat field")
    # two bins, 300 + 325 values: the counting rank, with ties
    sv2 = np.where(rng.random((1, P, S)) < 0.5, -70.0, -69.9)
    exp = _median_filter_db(sv2[0], n, m)
    pooled, _ = ops.pool_sv(_dev(torch, sv2), 0, n, m, func="nanmedian")
    _close(pooled.cpu().numpy()[0], exp, 1e-12, "two values")


def test_pool_sv_median_window_wider_than_a_workgroup_sweeps_every_window(env):
    """2m+1 > 256 columns (or a window that does not fit the LDS ring): one workgroup per output sample, every window
    swept from memory (the round-1 kernel)."""
    from echopype_amd import _lib

    torch, ops = env
    sv, _ = _scene(1, 4, 40, 12, spikes=False, nan_frac=0.1)
    exp = _median_filter_db(sv[0], 1, 130)
    with _lib.launch_trace() as tr:
        pooled, _ = ops.pool_sv(_dev(torch, sv), 0, 1, 130, func="nanmedian")
    assert "pool_median_kernel" in tr.kernels
    _close(pooled.cpu().numpy()[0], exp, 1e-9, "wide window")


@pytest.mark.parametrize("same_rows", [False, True, "mixed"])
@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_pool_sv_value_running_sums_equal_window_sums(env, dtype, same_rows):
    """Value-window pooling: per-row double-double running sums (default) == summing every window (workspace
    NULL) == the oracle's triple loop; range vectors that differ from ping to ping (row-by-row path), that are
    identical (interval sums + sliding sum down the columns) or both in one call; NaN-padded tails of different
    lengths, a 140 dB spike and a +inf sample."""
    torch, ops = env
    rng = np.random.default_rng(8)
    C, P, S, n, dbin = 2, 40, 300, 4, 1.45  # not a multiple of the 0.3 m step: no window edge on a sample
    sv, depth = _scene(C, P, S, 12, step=0.3)
    if same_rows is False:
        depth = depth * (1 + 0.01 * rng.random((C, P, 1)))      # a different range vector per ping
    elif same_rows == "mixed":
        depth[1] = depth[1] * (1 + 0.01 * rng.random((P, 1)))   # channel 0: one vector (sliding path), channel 1: not
    depth[:, 7, S - 20:] = np.nan
    depth[0, 30, S - 55:] = np.nan                               # rows of different valid lengths
    sv[0, 30, S - 55:] = np.nan
    sv[:, 7, S - 20:] = np.nan
    sv[0, 10, 100] = 60.0
    sv[1, 20, 50] = np.inf
    sv, depth = sv.astype(dtype), depth.astype(dtype)
    svt, rgt = _dev(torch, sv), _dev(torch, depth)
    nvalid, bad = ops.range_rows_check(rgt)
    assert bad == 0
    lo, hi = ops.nanminmax(rgt)
    a, ma = ops.pool_sv_value(svt, rgt, nvalid, dbin, n, 2.0, lo, hi, threshold=6.0, running_sums=True)
    b, mb = ops.pool_sv_value(svt, rgt, nvalid, dbin, n, 2.0, lo, hi, threshold=6.0, running_sums=False)
    a, b = a.cpu().numpy(), b.cpu().numpy()
    np.testing.assert_array_equal(np.isnan(a), np.isnan(b))
    _close(a, b, 1e-12 if dtype == "float64" else 1e-5, "running sums vs window sums")
    assert np.isposinf(a).any() and np.isfinite(a).any()
    exp = omask.pool_Sv(sv.astype(np.float64), depth.astype(np.float64), np.nanmean, dbin, n, 2.0)
    _close(a, exp, RTOL[dtype], "vs oracle")
    agree = ma.cpu().numpy() == mb.cpu().numpy()
    with np.errstate(invalid="ignore"):
        sure = ~(np.abs(sv.astype(np.float64) - exp - 6.0) < MARGIN[dtype])
    assert agree[sure].all()


@pytest.mark.parametrize("same_rows", [False, True, "mixed"])
@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_pool_sv_value_median_carried_window_equals_windows_from_memory(env, dtype, same_rows):
    """Value-window nanmedian: the window carried from ping to ping (channels with one range vector; workspace given)
    == every window taken from memory (workspace NULL) == the oracle's triple loop.  Rows of different valid lengths,
    a spike, +inf, pings near both ends (p - n < 0, p + n == P is pooled, p + n > P is not)."""
    from echopype_amd import _lib

    torch, ops = env
    rng = np.random.default_rng(9)
    C, P, S, n, dbin = 2, 40, 300, 4, 1.45
    sv, depth = _scene(C, P, S, 13, step=0.3)
    if same_rows is False:
        depth = depth * (1 + 0.01 * rng.random((C, P, 1)))
    elif same_rows == "mixed":
        depth[1] = depth[1] * (1 + 0.01 * rng.random((P, 1)))
    depth[:, 7, S - 20:] = np.nan
    depth[0, 30, S - 55:] = np.nan
    sv[0, 30, S - 55:] = np.nan
    sv[:, 7, S - 20:] = np.nan
    sv[0, 10, 100] = 60.0
    sv[1, 20, 50] = np.inf
    sv, depth = sv.astype(dtype), depth.astype(dtype)
    svt, rgt = _dev(torch, sv), _dev(torch, depth)
    nvalid, bad = ops.range_rows_check(rgt)
    assert bad == 0
    lo, hi = ops.nanminmax(rgt)
    with _lib.launch_trace() as tr:
        a, ma = ops.pool_sv_value(svt, rgt, nvalid, dbin, n, 2.0, lo, hi, func="nanmedian", threshold=6.0)
    assert "pool_value_median_slide_kernel" in tr.kernels
    with _lib.launch_trace() as tr:
        b, mb = ops.pool_sv_value(svt, rgt, nvalid, dbin, n, 2.0, lo, hi, func="nanmedian", threshold=6.0,
                                  running_sums=False)
    assert "pool_value_median_kernel" in tr.kernels
    a, b = a.cpu().numpy(), b.cpu().numpy()
    np.testing.assert_array_equal(a, b)   # the same two middle values, the same conversion
    np.testing.assert_array_equal(ma.cpu().numpy(), mb.cpu().numpy())
    exp = omask.pool_Sv(sv.astype(np.float64), depth.astype(np.float64), np.nanmedian, dbin, n, 2.0)
    if dtype == "float64":
        _close(a, exp, RTOL[dtype], "vs oracle")
    else:  # d +- bin in fp32 moves a sample on a window edge in or out: a median may jump to its neighbour value
        np.testing.assert_array_equal(np.isnan(a), np.isnan(exp))
        fin = np.isfinite(exp)
        off = np.abs(a[fin] - exp[fin]) > 1e-3 * np.maximum(np.abs(exp[fin]), 1.0)
        assert off.mean() < 2e-3 and np.abs(a[fin] - exp[fin])[off].max(initial=0.0) < 1.0
    assert np.isfinite(exp[:, P - n]).any() and np.isnan(exp[:, P - n + 1:]).all() and np.isnan(exp[:, :n]).all()


def test_pool_sv_value_median_segments_and_flat_field(env):
    """600 pings (two 512-ping segments) of one range vector; a stretch of identical values (more than 64 candidates in
    the median's bin: the radix selection over the window in memory) and an all-NaN stretch."""
    torch, ops = env
    rng = np.random.default_rng(10)
    C, P, S, n, dbin = 1, 600, 36, 3, 1.0
    sv = -75 + 3 * rng.standard_normal((C, P, S))
    sv[0, 100:140, :] = -70.0
    sv[0, 300:320, :] = np.nan
    sv[0, rng.random((P, S)) < 0.05] = np.nan
    depth = np.broadcast_to(2.0 + 0.25 * np.arange(S), (C, P, S)).copy()
    svt, rgt = _dev(torch, sv), _dev(torch, depth)
    nvalid, bad = ops.range_rows_check(rgt)
    lo, hi = ops.nanminmax(rgt)
    a, _ = ops.pool_sv_value(svt, rgt, nvalid, dbin, n, 3.0, lo, hi, func="nanmedian")
    exp = omask.pool_Sv(sv, depth, np.nanmedian, dbin, n, 3.0)
    _close(a.cpu().numpy(), exp, 1e-9, "600 pings")
    assert np.isnan(exp[0, 305:315, 10]).all() and np.isfinite(exp[0, 500:520, 10]).all()


def test_pool_sv_value_rows_longer_than_the_lds_copy(env):
    """8200 samples per ping: the running sums of a row no longer fit the fused kernel's LDS copy, the unfused
    kernels (running sums and interval sums through the workspace) take over -- same answers."""
    torch, ops = env
    rng = np.random.default_rng(17)
    C, P, S, n, dbin = 1, 14, 8200, 2, 2.2
    depth = (1.0 + 0.05 * np.arange(S))[None, None, :] + np.zeros((C, P, 1))
    sv = -75 + 5 * rng.standard_normal((C, P, S))
    sv[rng.random((C, P, S)) < 0.05] = np.nan
    svt, rgt = _dev(torch, sv), _dev(torch, depth)
    nvalid, bad = ops.range_rows_check(rgt)
    assert bad == 0
    lo, hi = ops.nanminmax(rgt)
    a, _ = ops.pool_sv_value(svt, rgt, nvalid, dbin, n, 5.0, lo, hi, running_sums=True)
    b, _ = ops.pool_sv_value(svt, rgt, nvalid, dbin, n, 5.0, lo, hi, running_sums=False)
    a, b = a.cpu().numpy(), b.cpu().numpy()
    np.testing.assert_array_equal(np.isnan(a), np.isnan(b))
    _close(a, b, 1e-12, "unfused running sums vs window sums")
    assert np.isfinite(a).any()


@pytest.mark.parametrize("case", ["blocks_of_pings", "span_beyond_the_lds_copy", "span_in_the_third_slot",
                                  "more_neighbours_than_span_slots", "rows_not_affine", "a_shallow_neighbour_row"])
@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_pool_sv_value_staged_neighbour_rows(env, dtype, case):
    """The LDS-staged value-window kernel (pings whose range rows differ) against summing every window: the range
    vector changing every 5 pings (the 8-ping groups straddle the changes: shared and separate intervals in one group),
    depth windows whose index span exceeds the LDS copy (direct loads for those pairs), a ping window of more
    neighbours than the kernel keeps spans for (direct loads for the whole group), and range rows that are monotone
    but far from affine in the sample index (the position guessed from the span's ends is wrong: searched instead);
    P, S not multiples of 8 / 256."""
    torch, ops = env
    rng = np.random.default_rng(23)
    if case in ("blocks_of_pings", "rows_not_affine", "a_shallow_neighbour_row"):
        C, P, S, n, dbin, step = 2, 43, 700, 6, 3.1, 0.3
    elif case == "span_beyond_the_lds_copy":
        C, P, S, n, dbin, step = 1, 21, 1500, 3, 85.0, 0.3
    elif case == "span_in_the_third_slot":  # 512 .. 767 samples: the lean kernel reads that slot when it stores the span
        C, P, S, n, dbin, step = 1, 21, 1500, 3, 50.0, 0.3
    else:
        C, P, S, n, dbin, step = 1, 1100, 70, 530, 1.3, 0.3
    sv, _ = _scene(C, P, S, 12, step=step)
    block = 5 if case == "blocks_of_pings" else 1
    scale = 1 + 0.01 * rng.random((C, (P + block - 1) // block, 1))
    depth = (1.5 + step * np.arange(S))[None, None, :] * np.repeat(scale, block, axis=1)[:, :P]
    if case == "rows_not_affine":
        k = np.arange(S)  # uneven steps (0.06 .. 0.54 m), a 3 m jump, a curvature, a per-ping offset
        uneven = np.cumsum(step * (0.2 + 1.6 * rng.random(S)))
        depth = (1.5 + uneven + 2e-4 * k * k + 3.0 * (k > 300))[None, None, :] * np.repeat(scale, block, axis=1)[:, :P] \
            + 0.7 * rng.random((C, P, 1))
        assert (np.diff(depth, axis=-1) > 0).all()
    if case == "a_shallow_neighbour_row":  # pings whose deepest sample lies above the deeper bands of their neighbours:
        depth[:, 20] *= 0.3                # the span of such a row is empty and starts at the row's end (kmin = S)
        depth[0, 31] *= 0.05
    depth[:, 7, S - 20:] = np.nan
    sv[:, 7, S - 20:] = np.nan
    sv[0, 10, 100 % S] = 60.0
    sv[0, 15, 50] = np.inf
    sv[rng.random((C, P, S)) < 0.03] = np.nan
    sv, depth = sv.astype(dtype), depth.astype(dtype)
    svt, rgt = _dev(torch, sv), _dev(torch, depth)
    nvalid, bad = ops.range_rows_check(rgt)
    assert bad == 0
    lo, hi = ops.nanminmax(rgt)
    a, ma = ops.pool_sv_value(svt, rgt, nvalid, dbin, n, 2.0, lo, hi, threshold=6.0, running_sums=True)
    b, mb = ops.pool_sv_value(svt, rgt, nvalid, dbin, n, 2.0, lo, hi, threshold=6.0, running_sums=False)
    a, b = a.cpu().numpy(), b.cpu().numpy()
    np.testing.assert_array_equal(np.isnan(a), np.isnan(b))
    _close(a, b, 1e-12 if dtype == "float64" else 1e-5, "staged running sums vs window sums")
    assert np.isposinf(a).any() and np.isfinite(a).any()
    if case in ("blocks_of_pings", "rows_not_affine", "a_shallow_neighbour_row") and dtype == "float64":  # (fp32 window edges d -+ bin round differently: a vs b only)
        exp = omask.pool_Sv(sv.astype(np.float64), depth.astype(np.float64), np.nanmean, dbin, n, 2.0)
        _close(a, exp, RTOL[dtype], "vs oracle")


@pytest.mark.parametrize("dtype", ["float64", "float32"])
@pytest.mark.parametrize("case", ["runs", "bridged", "short_runs", "one_side_ping"])
def test_pool_sv_value_runs_of_one_range_vector(env, dtype, case):
    """Round 6: inside a channel whose pings do not all share a range vector, the pings of a RUN that does take the
    sliding route when their ping window lies inside the run (value_slide_runs_kernel), the others the staged kernels --
    the same pooled values as summing every window (and the oracle's triple loop, clean/utils.py:29-106).  Runs of 60,
    23 and 90 pings with n = 9 (the 23-ping run has 5 eligible pings), NaN tails of different lengths inside a run, a
    row without a valid sample, a +inf sample; ``bridged``: a short row between two different vectors that agrees with
    both on its own samples -- the proposal joins the runs, the verification against the reference rejects the run;
    ``short_runs``: no run reaches 2 n + 1 pings; ``one_side_ping``: n = 1."""
    from echopype_amd import _lib
    torch, ops = env
    rng = np.random.default_rng(41)
    C, S, dbin, step = 2, 520, 2.3, 0.3
    n = 1 if case == "one_side_ping" else 9
    lens = [60, 23, 90, 40] if case != "short_runs" else [11, 7, 15, 9, 13, 12, 10, 14, 8, 16, 18, 17, 5, 18]
    P = sum(lens)
    sv, _ = _scene(C, P, S, 12, step=step)
    scale = np.repeat(1 + 0.004 * np.arange(len(lens)), lens)
    depth = (1.5 + step * np.arange(S))[None, None, :] * scale[None, :, None] * np.ones((C, 1, 1))
    depth[1] *= 1.0 + 0.0007 * rng.random((P, 1))  # channel 1: a vector of its own at every ping
    depth[0, 7, S - 20:] = np.nan
    depth[0, 30, S - 55:] = np.nan
    depth[0, 100, :] = np.nan                      # a row without a valid sample inside a run
    if case == "bridged":                          # ping 59 | 60: a row of 3 samples that equals BOTH neighbours there
        depth[0, 60:83] = depth[0, 0] + 0.0        # the second run takes the first run's vector ...
        depth[0, 61:83, 10:] += 0.004 * (np.arange(S - 10) + 1)  # ... except beyond sample 10, and row 60 is cut before
        depth[0, 60, 3:] = np.nan
    sv[np.isnan(depth)] = np.nan
    sv[0, 10, 100] = 60.0
    sv[0, 75, 50] = np.inf
    sv[rng.random((C, P, S)) < 0.03] = np.nan
    sv, depth = sv.astype(dtype), depth.astype(dtype)
    svt, rgt = _dev(torch, sv), _dev(torch, depth)
    nvalid, bad = ops.range_rows_check(rgt)
    assert bad == 0
    lo, hi = ops.nanminmax(rgt)
    with _lib.launch_trace() as tr:
        a, ma = ops.pool_sv_value(svt, rgt, nvalid, dbin, n, 2.0, lo, hi, threshold=6.0, running_sums=True)
    assert "value_slide_runs_kernel" in tr.kernels and "run_verify_kernel" in tr.kernels
    b, mb = ops.pool_sv_value(svt, rgt, nvalid, dbin, n, 2.0, lo, hi, threshold=6.0, running_sums=False)
    a, b = a.cpu().numpy(), b.cpu().numpy()
    np.testing.assert_array_equal(np.isnan(a), np.isnan(b))
    _close(a, b, 1e-12 if dtype == "float64" else 1e-5, "runs vs window sums")
    assert np.isposinf(a).any() and np.isfinite(a).any()
    if dtype == "float64" and case in ("runs", "bridged"):  # (the triple loop takes half a minute at this size)
        exp = omask.pool_Sv(sv.astype(np.float64), depth.astype(np.float64), np.nanmean, dbin, n, 2.0)
        _close(a, exp, RTOL[dtype], "vs oracle")


def test_pool_sv_everything_above_exclusion(env):
    torch, ops = env
    sv, _ = _scene(1, 6, 10, 3)
    pooled, mask = ops.pool_sv(_dev(torch, sv), 10, 2, 2)
    assert np.isnan(pooled.cpu().numpy()).all() and not mask.cpu().numpy().any()


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_apply_mask_and_mask_and(env, dtype):
    torch, ops = env
    rng = np.random.default_rng(4)
    src = rng.standard_normal((3, 7, 11)).astype(dtype)
    m1 = rng.random((3, 7, 11)) < 0.6
    m2 = rng.random((7, 11)) < 0.7  # channel-less mask, broadcast
    both = ops.mask_and(_dev(torch, m1.astype(np.uint8)), _dev(torch, m2.astype(np.uint8)))
    np.testing.assert_array_equal(both.cpu().numpy().astype(bool), m1 & m2[None])
    got = ops.apply_mask(_dev(torch, src), both).cpu().numpy()
    np.testing.assert_array_equal(got, omask.apply_mask(src, [m1, m2]).astype(dtype))
    got = ops.apply_mask(_dev(torch, src), _dev(torch, m2.astype(np.uint8)), fill_value=-999.0).cpu().numpy()
    np.testing.assert_array_equal(got, omask.apply_mask(src, m2, -999.0).astype(dtype))
    fill = rng.standard_normal((7, 11)).astype(dtype)
    got = ops.apply_mask(_dev(torch, src), _dev(torch, m1.astype(np.uint8)), fill_array=_dev(torch, fill))
    np.testing.assert_array_equal(got.cpu().numpy(), np.where(m1, src, fill[None]))


@pytest.mark.parametrize("dtype", ["float64", "float32"])
@pytest.mark.parametrize("shape", [(3, 7, 11), (4, 300, 2050)])
def test_apply_masks_is_the_and_the_selection_and_the_range_in_one_sweep(env, dtype, shape):
    """ops.apply_masks (epa_apply_masks): where(m1 & m2 & ..., src, fill) for up to four masks, channel-less ones broadcast,
    scalar or array fill, + the NaN-skipping {min, max} of the result -- against the oracle's apply_mask (mask/api.py:402-432)
    and against the separate kernels it replaces on the API's route (mask_and + apply_mask + nanminmax)."""
    torch, ops = env
    rng = np.random.default_rng(41)
    C, P, S = shape
    src = rng.standard_normal(shape).astype(dtype)
    src[rng.random(shape) < 0.05] = np.nan
    m1 = rng.random(shape) < 0.6
    m2 = rng.random((P, S)) < 0.7      # channel-less, broadcast
    m3 = rng.random(shape) < 0.9
    m4 = rng.random((P, S)) < 0.95
    dev = lambda m: _dev(torch, m.astype(np.uint8))
    for ms in ([m1], [m1, m2], [m1, m2, m3], [m1, m3, m2, m4]):
        out, mm = ops.apply_masks(_dev(torch, src), [dev(m) for m in ms], fill_value=-999.0, want_minmax=True)
        exp = omask.apply_mask(src, ms, -999.0).astype(dtype)
        np.testing.assert_array_equal(out.cpu().numpy(), exp)
        lo, hi = mm.cpu().tolist()
        assert lo == float(np.nanmin(exp)) and hi == float(np.nanmax(exp))
    # the separate kernels give the same array
    both = ops.mask_and(ops.mask_and(dev(m1), dev(m2)), dev(m3))
    sep = ops.apply_mask(_dev(torch, src), both, fill_value=-999.0)
    one, none = ops.apply_masks(_dev(torch, src), [dev(m1), dev(m2), dev(m3)], fill_value=-999.0)
    assert none is None and torch.equal(sep.view(torch.int64 if dtype == "float64" else torch.int32),
                                         one.view(torch.int64 if dtype == "float64" else torch.int32))
    # an array fill (NaN where it is NaN), and nothing kept / nothing a number -> {NaN, NaN}
    fill = rng.standard_normal((P, S)).astype(dtype)
    out, mm = ops.apply_masks(_dev(torch, src), [dev(m1), dev(m2)], fill_array=_dev(torch, fill), want_minmax=True)
    exp = np.where(m1 & m2[None], src, fill[None])
    np.testing.assert_array_equal(out.cpu().numpy(), exp)
    assert mm.cpu().tolist() == [float(np.nanmin(exp)), float(np.nanmax(exp))]
    out, mm = ops.apply_masks(_dev(torch, src), [dev(np.zeros(shape, bool))], want_minmax=True)
    assert np.isnan(out.cpu().numpy()).all() and all(np.isnan(v) for v in mm.cpu().tolist())
    with pytest.raises(ValueError):
        ops.apply_masks(_dev(torch, src), [dev(m1)] * 5)


def _box_mean_db(sv, n, m, s0):
    """Index-window pooled Sv by separable running sums in extended precision (np.pad 'symmetric' == scipy
    'reflect', periodic for windows wider than the data)."""
    if s0 >= sv.shape[1]:
        return np.full(sv.shape, np.nan)
    # (cumulative sums cancel: good to ~1e-9 dB only while the window is not ~1e9 times smaller than the row sum)
    lin = (10 ** (np.longdouble(sv[:, s0:]) / 10))
    ok = ~np.isnan(lin)
    val = np.where(ok, lin, np.longdouble(0))

    def box(a, k, axis):
        a = np.moveaxis(a, axis, -1)
        L = a.shape[-1]
        idx = np.arange(-k, L + k)
        per = 2 * L
        idx = np.mod(idx, per)
        idx = np.where(idx < L, idx, per - 1 - idx)
        ext = np.take(a, idx, axis=-1)
        cs = np.concatenate([np.zeros(a.shape[:-1] + (1,), a.dtype), np.cumsum(ext, axis=-1)], axis=-1)
        out = cs[..., 2 * k + 1:] - cs[..., :-(2 * k + 1)]
        return np.moveaxis(out, -1, axis)

    s = box(box(val, m, 1), n, 0)
    c = box(box(ok.astype(np.int64), m, 1), n, 0)
    with np.errstate(all="ignore"):
        out = np.where(c > 0, 10 * np.log10(s / np.maximum(c, 1)), np.nan).astype(np.float64)
    full = np.full(sv.shape, np.nan)
    full[:, s0:] = out
    return full


@pytest.mark.parametrize("P,S,n,m,s0", [
    (700, 1100, 25, 53, 100),    # two ping segments, two range tiles (1024 + ragged), block-scan range pass
    (513, 2300, 1, 4, 0),        # w = 9: smallest scanned window; three range tiles
    (40, 1500, 600, 255, 7),     # w = 511: largest scanned window; ping window wraps the data many times
    (90, 1300, 3, 256, 0),       # w = 513: grouped range kernel
    (33, 60, 0, 3, 59),          # w = 7: grouped kernel; a single pooled column
    (1030, 35, 2, 20, 3),        # window wider than the row (reflect wraps), three ping segments
])
def test_pool_sv_index_windows_against_running_sums(env, P, S, n, m, s0):
    torch, ops = env
    rng = np.random.default_rng(P + S + n + m)
    sv = -80 + 8 * rng.standard_normal((1, P, S))
    sv[0, rng.random((P, S)) < 0.03] += 50
    sv[0, rng.random((P, S)) < 0.05] = np.nan
    sv[0, P // 2, :] = np.nan
    exp = _box_mean_db(sv[0], n, m, s0)
    pooled, mask = ops.pool_sv(_dev(torch, sv), s0, n, m, threshold=9.0)
    got = pooled.cpu().numpy()[0]
    np.testing.assert_array_equal(np.isnan(got), np.isnan(exp))
    _close(got, exp, 1e-9, f"pooled {P}x{S} window {2*n+1}x{2*m+1}")
    with np.errstate(invalid="ignore"):
        margin = sv[0] - exp - 9.0
    sure = ~(np.abs(margin) < 1e-7)
    np.testing.assert_array_equal(mask.cpu().numpy()[0].astype(bool)[sure], (margin > 0)[sure])
