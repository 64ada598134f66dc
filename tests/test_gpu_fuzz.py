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
