"""Randomised shape / flag sweep of the core path through the Dataset API against the oracle: odd
and tiny shapes (S not a multiple of 4, a single ping, a single channel), random NaN patterns, every
(skipna, closed) combination, bins finer and coarser than the data, noise blocks larger than the
array -- the combinations that pick between the specialised and the generic kernels."""
import numpy as np
import pytest

import oracle_chain as oc
from oracle import clean as oclean
from oracle import commongrid as ogrid
from test_gpu_api import close

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ep():
    import torch

    if not torch.cuda.is_available():
        pytest.fail("these tests need a GPU")
    import echopype_amd

    return echopype_amd


def _case(seed):
    rng = np.random.default_rng(1000 + seed)
    C = int(rng.integers(1, 4))
    P = int(rng.choice([1, 2, 3, 7, 20, 41, 97]))
    S = int(rng.choice([1, 3, 4, 5, 17, 64, 130, 255, 256, 1023, 1030]))
    return rng, C, P, S


@pytest.mark.parametrize("seed", range(60))
def test_random_shapes_and_flags(ep, seed):
    rng, C, P, S = _case(seed)
    d = ep.synth.ek60_numpy(C, P, S, seed=seed, vary_tau=bool(rng.integers(0, 2)))
    raw = d["backscatter_r"]
    raw[rng.random(raw.shape) < 0.05] = np.nan
    if P > 2 and rng.random() < 0.5:
        raw[:, int(rng.integers(0, P))] = np.nan  # a whole ping missing
    dtype = "float64"
    ed = ep.echodata.from_ek60_arrays(d)
    skipna, closed = bool(rng.integers(0, 2)), str(rng.choice(["left", "right"]))
    rbin = str(rng.choice(["0.3m", "1m", "7.5m", "500m"]))
    tbin = str(rng.choice(["1s", "7s", "20s", "3min"]))
    sv, er = oc.ek60(d, "Sv")
    ds = ep.calibrate.compute_Sv(ed, dtype=dtype)
    close(ds["Sv"].values, sv, 1e-9, f"Sv C={C} P={P} S={S}")
    np.testing.assert_array_equal(ds["echo_range"].values, er)
    if not np.isfinite(er).any():
        return
    exp_mv, t_left, r_left = ogrid.compute_MVBS(sv, er, d["ping_time"], rbin, tbin, skipna=skipna, closed=closed)
    mv = ep.commongrid.compute_MVBS(ds, range_bin=rbin, ping_time_bin=tbin, skipna=skipna, closed=closed)
    close(mv["Sv"].values, exp_mv, 1e-9, f"MVBS {rbin} {tbin} skipna={skipna} closed={closed}")
    np.testing.assert_array_equal(mv["ping_time"].values, t_left)
    if skipna and closed == "left":
        ds2, mv2 = ep.compute_Sv_MVBS(ed, range_bin=rbin, ping_time_bin=tbin)
        close(mv2["Sv"].values, exp_mv, 1e-9, "fused MVBS")
        close(ds2["Sv"].values, sv, 1e-9, "fused Sv")
    ping_num, rsn = int(rng.choice([1, 3, 20, 500])), int(rng.choice([1, 5, 50, 5000]))
    nmax = None if rng.random() < 0.5 else "-110.0dB"
    exp_n, exp_c = oclean.remove_background_noise(sv, er, d["absorption_indicative"], ping_num, rsn, nmax, "3.0dB")
    ep.clean.remove_background_noise(ds, ping_num, rsn, background_noise_max=nmax)
    close(ds["Sv_noise"].values, exp_n, 1e-9, f"Sv_noise {ping_num}x{rsn}")
    close(ds["Sv_corrected"].values, exp_c, 1e-7, "Sv_corrected")
    ds3, mv3 = ep.compute_Sv_clean_MVBS(ed, ping_num, rsn, background_noise_max=nmax, range_bin=rbin,
                                        ping_time_bin=tbin, skipna=skipna, closed=closed)
    close(ds3["Sv_corrected"].values, exp_c, 1e-7, "two-pass Sv_corrected")
    exp_mvc, _, _ = ogrid.compute_MVBS(exp_c, er, d["ping_time"], rbin, tbin, skipna=skipna, closed=closed)
    close(mv3["Sv"].values, exp_mvc, 1e-7, "two-pass MVBS")


@pytest.mark.parametrize("seed", range(30))
def test_random_masks_and_nasc(ep, seed):
    """Noise masks, apply_mask and NASC on random small shapes / window sizes (windows larger than the data,
    a single ping, odd lengths) against the oracle."""
    from oracle import masks as omask
    from oracle import nasc as onasc
    from test_gpu_masks import _scene
    from test_gpu_masks_api import _ds

    rng = np.random.default_rng(5000 + seed)
    C, P, S = int(rng.integers(1, 4)), int(rng.choice([1, 2, 5, 13, 32])), int(rng.choice([2, 7, 30, 65, 128]))
    sv, depth = _scene(C, P, S, seed, step=float(rng.choice([0.2, 0.5, 1.3])), spikes=P > 2 and S > 8)
    ds = _ds(ep, sv, depth)
    n = int(rng.choice([1, 2, 6]))
    dbin = str(rng.choice(["1m", "2.5m", "40m"]))
    for index in (True, False):
        m = ep.clean.mask_impulse_noise(ds, depth_bin=dbin, num_side_pings=n, use_index_binning=index).values
        np.testing.assert_array_equal(m, omask.mask_impulse_noise(sv, depth, dbin, n, "10.0dB", index))
    excl = f"{float(rng.choice([0.0, 3.0, 1e4]))}m"
    func = str(rng.choice(["nanmean", "nanmedian"]))
    m = ep.clean.mask_transient_noise(ds, func=func, depth_bin=dbin, num_side_pings=n, exclude_above=excl,
                                      transient_noise_threshold="6.0dB", use_index_binning=True).values
    f = np.nanmean if func == "nanmean" else np.nanmedian
    m_c = omask.nsamples_per_bin(depth, float(dbin[:-1]))
    # scipy's generic_filter (the oracle's engine) stops reflecting correctly once the window is several
    # times the array (checked against np.pad(mode="symmetric"), which the HIP kernel matches); the
    # reference's dask_image path cannot take such windows either
    if 2 * m_c.max() + 1 <= 2 * S and 2 * n + 1 <= 2 * P:
        pooled = omask.index_binning_pool_Sv(sv, depth, f, float(dbin[:-1]), n, float(excl[:-1]))
        with np.errstate(invalid="ignore"):
            margin = sv - pooled - 6.0
        sure = ~(np.abs(margin) < 1e-9)
        np.testing.assert_array_equal(m[sure], (margin > 0)[sure])
    up, lw = sorted(rng.uniform(np.nanmin(depth), np.nanmax(depth), 2))
    att = ep.clean.mask_attenuated_signal(ds, upper_limit_sl=f"{up:09.3f}m", lower_limit_sl=f"{lw:09.3f}m",
                                          num_side_pings=n, attenuation_signal_threshold="-3.0dB").values
    exp_att = omask.mask_attenuated_signal(sv, depth, f"{up:09.3f}m", f"{lw:09.3f}m", n, "-3.0dB")
    np.testing.assert_array_equal(att, exp_att)
    out = ep.mask.apply_mask(ds, [ep.DataArray(~att, DIMS3), ep.DataArray(~m, DIMS3)], fill_value=-1.0)
    np.testing.assert_array_equal(out["Sv"].values, omask.apply_mask(sv, [~exp_att, ~m], -1.0))
    # NASC on the same scene with a random track
    lat = 40.0 + np.cumsum(rng.uniform(0, 2e-4, P))
    lon = -125.0 + np.cumsum(rng.uniform(0, 2e-4, P))
    ds["latitude"], ds["longitude"] = (("ping_time",), lat), (("ping_time",), lon)
    if P > 1:
        closed = str(rng.choice(["left", "right"]))
        got = ep.commongrid.compute_NASC(ds, range_bin=dbin, dist_bin="0.01nmi", closed=closed)
        exp = onasc.compute_NASC(sv, depth, lat, lon, ds["ping_time"].values, float(dbin[:-1]), 0.01, closed=closed)
        g, e = got["NASC"].values, exp["NASC"]
        np.testing.assert_array_equal(np.isnan(g), np.isnan(e))
        np.testing.assert_allclose(g[~np.isnan(e)], e[~np.isnan(e)], rtol=1e-10)


DIMS3 = ("channel", "ping_time", "range_sample")


@pytest.mark.parametrize("seed", range(12))
def test_random_ek80_complex(ep, seed):
    """EK80 complex (BB through the LDS-FFT path over one to several tiles, CW) on random shapes, NaN
    patterns and output types against the oracle, judged like test_compute_Sv_ek80_bb."""
    from test_gpu_api import _ek80

    rng = np.random.default_rng(9000 + seed)
    wf = "BB" if seed % 3 else "CW"
    P, S = int(rng.choice([1, 3, 6])), int(rng.choice([300, 1200, 1872, 1873, 2500, 4100]))
    dtype = str(rng.choice(["float64", "float32"]))
    d, filt = _ek80(ep, wf, C=2, P=P, S=S, mixed_nan=bool(rng.integers(0, 2)), seed=seed)
    cal = str(rng.choice(["Sv", "TS"]))
    fn = ep.calibrate.compute_Sv if cal == "Sv" else ep.calibrate.compute_TS
    ds = fn(ep.echodata.from_ek80_arrays(d, filt), waveform_mode=wf, encode_mode="complex", dtype=dtype)
    (exp, exp_r, prx), _ = oc.ek80_complex(d, filt, cal)
    got = ds[cal].values.astype(np.float64)
    np.testing.assert_array_equal(np.isnan(got), np.isnan(exp))
    if wf == "CW":
        close(got, exp, 1e-9 if dtype == "float64" else 1e-3, f"CW {cal}")
    else:
        from bb_tolerance import assert_bb_close

        assert_bb_close(got, exp, dtype, prx=prx)
    if dtype == "float64":
        np.testing.assert_array_equal(ds["echo_range"].values, exp_r)


@pytest.mark.parametrize("seed", range(24))
def test_random_ping_times(ep, seed):
    """compute_MVBS on irregular ping cadences: random gaps (empty time bins), bins crossing the midnight
    anchor, duplicate timestamps, unsorted pings and NaT -- the time-bin edges come from pandas' resample in
    the oracle, from the host arithmetic of commongrid/utils.py here."""
    rng = np.random.default_rng(7000 + seed)
    C, P, S = 2, int(rng.choice([5, 40, 160])), int(rng.choice([16, 200]))
    d = ep.synth.ek60_numpy(C, P, S, seed=seed)
    start = np.datetime64("2026-05-01T23:58:30", "ns") + np.timedelta64(int(rng.integers(0, 3600)), "s")
    gaps = rng.choice([0.0, 0.2, 1.0, 3.7, 45.0, 400.0], size=P, p=[0.05, 0.3, 0.4, 0.15, 0.07, 0.03])
    t = start + (np.cumsum(gaps) * 1e9).astype("timedelta64[ns]")
    if seed % 3 == 1:
        t = t[rng.permutation(P)]  # unsorted
    sv, er = oc.ek60({**d, "ping_time": t}, "Sv")
    ds = ep.Dataset(coords={"channel": d["channel"], "ping_time": t, "range_sample": np.arange(S)})
    dims = ("channel", "ping_time", "range_sample")
    ds["Sv"], ds["echo_range"] = (dims, sv), (dims, er)
    tbin = str(rng.choice(["2s", "20s", "1min", "10min", "1h"]))
    closed = str(rng.choice(["left", "right"]))
    skipna = bool(rng.integers(0, 2))
    exp, t_left, r_left = ogrid.compute_MVBS(sv, er, t, "5m", tbin, skipna=skipna, closed=closed)
    mv = ep.commongrid.compute_MVBS(ds, range_bin="5m", ping_time_bin=tbin, skipna=skipna, closed=closed)
    np.testing.assert_array_equal(mv["ping_time"].values, t_left)
    np.testing.assert_array_equal(mv["echo_range"].values, r_left)
    close(mv["Sv"].values, exp, 1e-9, f"{tbin} closed={closed} skipna={skipna}")


@pytest.mark.parametrize("seed", range(30))
def test_lazy_echo_range_paths_against_the_array_paths(ep, seed):
    """What the steps after compute_Sv make of a LAZY echo_range (coefficient rows, the raw samples' NaN pattern) against
    the same steps on the written array, over random shapes / dtypes / flags: compute_MVBS (same bins and members: sums
    agree to rounding), remove_background_noise (to the last-bit noise of the estimate's atomics), add_depth (bit-identical) and the variable itself when read."""
    rng, C, P, S = _case(100 + seed)
    S = max(S, 4)
    dtype = str(rng.choice(["float64", "float32"]))
    d = ep.synth.ek60_numpy(C, P, S, seed=seed, vary_tau=bool(rng.integers(0, 2)), ss_every=int(rng.choice([1, 3, 1000])))
    raw = d["backscatter_r"]
    raw[rng.random(raw.shape) < 0.05] = np.nan
    if P > 2 and rng.random() < 0.5:
        raw[:, int(rng.integers(0, P))] = np.nan
    ed = ep.echodata.from_ek60_arrays(d)
    closed = str(rng.choice(["left", "right"]))
    rbin = str(rng.choice(["0.3m", "1m", "7.5m", "500m"]))
    tbin = str(rng.choice(["1s", "7s", "20s", "3min"]))

    def fresh(written):
        ds = ep.calibrate.compute_Sv(ed, dtype=dtype)
        if written:
            ds["echo_range"] = ep.xr_lite.DataArray(ep.DeviceArray(ds["echo_range"].data.tensor), ds["echo_range"].dims,
                                                    attrs=ds["echo_range"].attrs)
        return ds

    lazy, arr = fresh(False), fresh(True)
    if not np.isfinite(arr["echo_range"].values).any():
        return
    tol = dict(rtol=1e-12, atol=1e-12) if dtype == "float64" else dict(rtol=1e-5, atol=1e-4)
    a = ep.commongrid.compute_MVBS(lazy, range_bin=rbin, ping_time_bin=tbin, closed=closed)
    b = ep.commongrid.compute_MVBS(arr, range_bin=rbin, ping_time_bin=tbin, closed=closed)
    np.testing.assert_array_equal(np.isnan(a["Sv"].values), np.isnan(b["Sv"].values))
    np.testing.assert_allclose(a["Sv"].values, b["Sv"].values, **tol)
    np.testing.assert_array_equal(a["echo_range"].values, b["echo_range"].values)
    pn, rsn = int(rng.choice([1, 5, 20, 300])), int(rng.choice([1, 4, 50, 5000]))
    snr = str(rng.choice(["0.0dB", "3.0dB"]))
    na = ep.clean.remove_background_noise(lazy, pn, rsn, SNR_threshold=snr)
    nb = ep.clean.remove_background_noise(arr, pn, rsn, SNR_threshold=snr)
    assert not getattr(lazy["echo_range"].data, "materialized", False)     # (odd S: K1 wrote the array, nothing is lazy)
    # (the noise estimate sums through LDS atomics: two runs may differ in the last bit, and with them whole ping blocks
    # of Sv_noise and the odd sample sitting exactly on the SNR threshold)
    ntol = dict(rtol=1e-13, atol=1e-12) if dtype == "float64" else dict(rtol=2e-6, atol=1e-4)
    np.testing.assert_array_equal(np.isnan(na["Sv_noise"].values), np.isnan(nb["Sv_noise"].values))
    np.testing.assert_allclose(na["Sv_noise"].values, nb["Sv_noise"].values, **ntol)
    ca, cb = na["Sv_corrected"].values, nb["Sv_corrected"].values
    both = np.isfinite(ca) & np.isfinite(cb)
    assert (np.isnan(ca) != np.isnan(cb)).mean() < 2e-3
    np.testing.assert_allclose(ca[both], cb[both], rtol=1e-9 if dtype == "float64" else 1e-3, atol=1e-9 if dtype == "float64" else 1e-4)
    off, tilt = float(rng.uniform(0, 9)), float(rng.uniform(0, 30))
    da = ep.consolidate.add_depth(lazy, depth_offset=off, tilt=tilt)
    db = ep.consolidate.add_depth(arr, depth_offset=off, tilt=tilt)
    assert not getattr(lazy["echo_range"].data, "materialized", False)     # (odd S: K1 wrote the array, nothing is lazy)
    np.testing.assert_array_equal(da["depth"].values, db["depth"].values)
    assert da["depth"].data.cached_stats() == db["depth"].data.cached_stats()
    np.testing.assert_array_equal(lazy["echo_range"].values, arr["echo_range"].values)


# ---- medium volumes that actually reach the specialised kernels (round 4) ---------------------------------------------------
# The planner (block_reduce.hip make_plan) sends time bins of more than 8 pings to a two-stage generic reduction unless
# C * n_tbins >= 1024: the tiny shapes above therefore never reach fused_sv_mvbs_kernel, mvbs_of_sv_fixed / rows_kernel or
# the chain's pass-2 kernels.  These cases do (asserted through the launch trace), with the sound-speed regimes that pick
# between them and the range lengths that leave a partial last wavefront.
@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("EPA_FUZZ_MEDIUM", "12"))))  # (a longer hunt: EPA_FUZZ_MEDIUM=100)
def test_random_medium_volumes_on_the_specialised_kernels(ep, seed):
    import logging

    from echopype_amd import _lib

    rng = np.random.default_rng(7000 + seed)
    C = int(rng.integers(1, 4))
    S = int(rng.choice([260, 1000, 1028, 2052, 4096]))
    P = int(rng.choice([1500, 2400, 3100]))
    regime = str(rng.choice(["constant", "steps", "every-ping", "jitter"]))
    d = ep.synth.ek60_numpy(C, P, S, seed=seed, ss_every={"constant": 10**9, "steps": 37, "every-ping": 1, "jitter": 1}[regime])
    if regime == "jitter":  # metres per second from ping to ping: columns cross range-bin edges inside a time bin
        d["sound_speed_indicative"] = d["sound_speed_indicative"] + 6.0 * rng.random((1, P))
    raw = d["backscatter_r"]
    raw[rng.random(raw.shape) < 0.01] = np.nan
    raw[:, int(rng.integers(0, P))] = np.nan
    tbin = str(rng.choice(["5s", "7s", "8s"]))            # <= 8 pings per bin: single-stage reduction
    rbin = str(rng.choice(["0.25m", "1m", "5m"]))
    sv, er = oc.ek60(d, "Sv")
    exp_mv, t_left, r_left = ogrid.compute_MVBS(sv, er, d["ping_time"], rbin, tbin)
    logging.disable(logging.WARNING)
    try:
        for resident in (False, True):
            ed = ep.echodata.from_ek60_arrays(d)
            if resident:
                ed.to_device()
            # (1) the two reference calls: Sv deferred, written by compute_MVBS's pass (statistics variant of the fused kernel)
            with _lib.launch_trace() as tr:
                ds = ep.calibrate.compute_Sv(ed)
                mv = ep.commongrid.compute_MVBS(ds, range_bin=rbin, ping_time_bin=tbin)
                shape = mv["Sv"].shape
            assert "fused_sv_mvbs_kernel" in tr.kernels, (tr.kernels, C, P, S, tbin)
            close(mv["Sv"].values, exp_mv, 1e-9, f"deferred MVBS {regime} C={C} P={P} S={S} {rbin} {tbin}")
            close(ds["Sv"].values, sv, 1e-9, "Sv written by compute_MVBS")
            np.testing.assert_array_equal(mv["echo_range"].values, r_left)
            np.testing.assert_array_equal(mv["ping_time"].values, t_left)
            assert shape == exp_mv.shape
        # (2) a second grid on the Sv array now there: the fixed-bin / per-row kernels through the coefficient rows
        rbin2 = "2m" if rbin != "5m" else "0.5m"
        exp2, _, _ = ogrid.compute_MVBS(sv, er, d["ping_time"], rbin2, tbin)
        with _lib.launch_trace() as tr:
            mv2 = ep.commongrid.compute_MVBS(ds, range_bin=rbin2, ping_time_bin=tbin)
        assert "mvbs_of_sv_fixed_kernel" in tr.kernels and "mvbs_of_sv_rows_kernel" in tr.kernels, tr.kernels
        close(mv2["Sv"].values, exp2, 1e-9, f"MVBS of the Sv array {regime} {rbin2}")
        # (3) the chain in two sweeps: uniform / drift / general pass 2 by regime
        pn, rsn = int(rng.choice([5, 20])), int(rng.choice([50, 130]))
        exp_n, exp_c = oclean.remove_background_noise(sv, er, d["absorption_indicative"], pn, rsn, None, "3.0dB")
        exp_mvc, _, _ = ogrid.compute_MVBS(exp_c, er, d["ping_time"], rbin, tbin)
        with _lib.launch_trace() as tr:
            ds3, mv3 = ep.compute_Sv_clean_MVBS(ed, pn, rsn, range_bin=rbin, ping_time_bin=tbin)
            mv3["Sv"].shape
        assert "sv_noise_fast_kernel" in tr.kernels and "sv_denoise_mvbs_uniform_kernel" in tr.kernels, tr.kernels
        close(ds3["Sv_noise"].values, exp_n, 1e-9, f"chain Sv_noise {regime}")
        close(ds3["Sv_corrected"].values, exp_c, 1e-7, f"chain Sv_corrected {regime}")
        close(mv3["Sv"].values, exp_mvc, 1e-7, f"chain MVBS {regime}")
    finally:
        logging.disable(logging.NOTSET)


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("EPA_FUZZ_CARRIED", "16"))))
def test_random_carried_windows_against_the_windows_taken_from_memory(ep, seed):
    """The round-4 carried-window kernels (nanmedian pooling by value windows, attenuated-signal mask) on random
    medium shapes -- ping counts around the 512-ping segments / the walk's chunks, layers and windows around the lane
    counts, NaN tails, flat stretches, quantised values (ties) -- against the kernels that take every window from
    memory (the workspace-free pooling; the per-ping attenuated kernel, reached with S % 4 != 0)."""
    import torch

    from echopype_amd import ops

    rng = np.random.default_rng(9000 + seed)
    C = int(rng.integers(1, 3))
    P = int(rng.choice([3, 40, 511, 513, 700, 1100]))
    S = 4 * int(rng.integers(6, 120))
    sv = -70 + float(rng.choice([0.5, 4.0])) * rng.standard_normal((C, P, S)) - 10 * np.linspace(0, 1, S)
    if rng.random() < 0.5:
        sv = np.round(sv * 4) / 4                       # quantised: many equal values
    att = rng.random((C, P)) < 0.08
    sv[att] -= rng.uniform(3, 40, size=int(att.sum()))[:, None]
    sv[rng.random((C, P, S)) < float(rng.choice([0.0, 0.05, 0.4]))] = np.nan
    if P > 60:
        sv[0, 20:45] = -71.0
        sv[C - 1, 50:55] = np.nan
    step = float(rng.choice([0.2, 0.5]))
    depth = np.broadcast_to(1.0 + step * np.arange(S), (C, P, S)).copy()
    kind = int(rng.integers(0, 3))
    if kind == 1:
        depth = depth + 2 * step * np.sin(np.arange(P) / 5.0)[None, :, None]     # heave
    elif kind == 2 and P > 4:
        depth[:, P // 2:] *= 1.07                                                  # limits change once
    tail = rng.random((C, P)) < 0.1
    depth[tail, S - S // 10:] = np.nan
    sv[np.isnan(depth)] = np.nan
    svt, rgt = torch.from_numpy(sv).cuda(), torch.from_numpy(depth).cuda()
    # attenuated-signal mask
    n = int(rng.choice([1, 2, 7, 15]))
    up, lw = sorted(rng.uniform(1.0, 1.0 + step * S, 2))
    pad = torch.nn.functional.pad
    got = ops.attenuated_mask(svt, rgt, float(up), float(lw), n, -4.0)
    ref = ops.attenuated_mask(pad(svt, (0, 1), value=float("nan")).contiguous(), pad(rgt, (0, 1), value=1.0e7).contiguous(),
                              float(up), float(lw), n, -4.0)[:, :, :S]
    got, ref = got.cpu().numpy(), ref.cpu().numpy()
    for c, p in np.argwhere(got[:, :, 0] != ref[:, :, 0]):
        # only a ping whose difference of medians sits ON the threshold may differ (quantised values: 10 log10(10^(x/10))
        # rounds differently where the compiler contracts the two kernels' multiplications differently)
        iu, il = int(np.argmin(np.abs(depth[c, p] - up))), int(np.argmin(np.abs(depth[c, p] - lw)))
        with np.errstate(invalid="ignore"):
            pm = 10 * np.log10(np.nanmedian(10 ** (sv[c, p, iu:il] / 10)))
            bm = 10 * np.log10(np.nanmedian(10 ** (sv[c, p - n:p + n, iu:il] / 10)))
        assert abs(pm - bm + 4.0) < 1e-9, f"attenuated seed {seed} ping {c},{p}: {pm} - {bm}"
    assert (got == got[:, :, :1]).all() and (got[:, :, 0] != ref[:, :, 0]).sum() <= max(4, 0.02 * C * P)
    # nanmedian pooling by value windows (one range vector per channel unless heave made them differ)
    nvalid, bad = ops.range_rows_check(rgt)
    assert bad == 0
    lo, hi = ops.nanminmax(rgt)
    ns = int(rng.choice([0, 1, 3, 9]))
    dbin = float(rng.choice([0.6, 2.3, 11.0]))
    excl = float(rng.choice([0.0, 5.0]))
    a, ma = ops.pool_sv_value(svt, rgt, nvalid, dbin, ns, excl, lo, hi, func="nanmedian", threshold=5.0)
    b, mb = ops.pool_sv_value(svt, rgt, nvalid, dbin, ns, excl, lo, hi, func="nanmedian", threshold=5.0,
                              running_sums=False)
    np.testing.assert_array_equal(a.cpu().numpy(), b.cpu().numpy(), err_msg=f"value median seed {seed}")
    np.testing.assert_array_equal(ma.cpu().numpy(), mb.cpu().numpy())
    # nanmedian pooling by index windows: a small cut against scipy's generic_filter
    import scipy.ndimage

    Pc, Sc = min(P, 60), min(S, 48)
    cut = sv[0, :Pc, :Sc]
    nn, mm = int(rng.integers(0, 5)), int(rng.integers(0, 9))
    if 2 * nn + 1 <= 2 * Pc and 2 * mm + 1 <= 2 * Sc:
        with np.errstate(invalid="ignore", divide="ignore"), __import__("warnings").catch_warnings():
            __import__("warnings").simplefilter("ignore", RuntimeWarning)
            exp = 10 * np.log10(scipy.ndimage.generic_filter(10 ** (cut / 10), np.nanmedian, size=[2 * nn + 1, 2 * mm + 1],
                                                             mode="reflect"))
        got, _ = ops.pool_sv(torch.from_numpy(np.ascontiguousarray(cut[None])).cuda(), 0, nn, mm, func="nanmedian")
        got = got.cpu().numpy()[0]
        np.testing.assert_array_equal(np.isnan(got), np.isnan(exp))
        fin = np.isfinite(exp)
        assert np.max(np.abs(got[fin] - exp[fin]) / np.maximum(np.abs(exp[fin]), 1.0), initial=0.0) < 1e-9
