"""Kernel-level parity: HIP kernels (through the C ABI, echopype_amd.ops) vs the CPU oracle on the
same seeded inputs.  Needs a real MI355X: `pytest -m gpu`.

Tolerances (BASELINE.json north_star): fp64 1e-5 relative, fp32 1e-3 relative.  The fp64 kernels
are in fact held to 1e-9 here (observed ~1e-13): a looser pass would hide a formula slip.
"""
import warnings

import numpy as np
import pytest

import kat_fixtures as kf
from oracle import calibrate as ocal
from oracle import clean as oclean
from oracle import commongrid as ogrid

pytestmark = pytest.mark.gpu

RTOL = {"float64": 1e-9, "float32": 1e-3}
NORTH_STAR_RTOL = {"float64": 1e-5, "float32": 1e-3}


@pytest.fixture(scope="module")
def env():
    import torch

    if not torch.cuda.is_available():
        pytest.fail("these tests need a GPU (run with -m 'not gpu' on CPU boxes)")
    from echopype_amd import ops, synth

    return torch, ops, synth


def _dev(torch, a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).cuda()
    return t.to(dtype) if dtype is not None else t


def _assert_close(got, exp, rtol, what=""):
    got, exp = np.asarray(got, dtype=np.float64), np.asarray(exp, dtype=np.float64)
    assert got.shape == exp.shape, (what, got.shape, exp.shape)
    np.testing.assert_array_equal(np.isnan(got), np.isnan(exp), err_msg=f"{what}: NaN pattern")
    fin = ~np.isnan(exp)
    np.testing.assert_array_equal(np.isinf(got[fin]), np.isinf(exp[fin]), err_msg=f"{what}: inf pattern")
    fin &= np.isfinite(exp)
    # relative to max(|expected|, 1): dB quantities cross zero, where a pure relative error is
    # meaningless; below 1 dB the bound is absolute (rtol dB)
    err = np.abs(got[fin] - exp[fin]) / np.maximum(np.abs(exp[fin]), 1.0)
    assert err.size == 0 or err.max() <= rtol, f"{what}: max rel err {err.max():.3e} > {rtol}"


def _oracle_ek60(d, cal_type):
    C, P, S = d["backscatter_r"].shape
    gain = ocal.vend_cal_params_power(d["transmit_duration_nominal"], d["pulse_length"], d["gain_correction"])
    sa = ocal.vend_cal_params_power(d["transmit_duration_nominal"], d["pulse_length"], d["sa_correction"])
    return ocal.cal_power_ek(
        d["backscatter_r"], sonar="EK60", cal_type=cal_type, sample_interval=d["sample_interval"],
        sound_speed=d["sound_speed_indicative"], absorption=d["absorption_indicative"],
        transmit_power=d["transmit_power"], tau_nominal=d["transmit_duration_nominal"], gain=gain,
        sa_correction=sa, psi=d["equivalent_beam_angle"], f_nominal=d["frequency_nominal"],
        tau_eff=d["transmit_duration_nominal"][:, 0])


def _coef_ek60(torch, ops, d, cal_type, sonar="EK60", gpt=None):
    f64 = torch.float64
    return ops.power_coef_ek(
        _dev(torch, d["sample_interval"], f64), _dev(torch, d["transmit_duration_nominal"], f64),
        _dev(torch, d["transmit_power"], f64), _dev(torch, d["sound_speed_indicative"], f64),
        _dev(torch, d["absorption_indicative"], f64), _dev(torch, d["gain_correction"], f64),
        _dev(torch, d["sa_correction"], f64), _dev(torch, d["equivalent_beam_angle"], f64),
        _dev(torch, d["frequency_nominal"], f64), _dev(torch, d["transmit_duration_nominal"][:, 0].copy(), f64),
        sonar=sonar, cal_type=cal_type, pulse_length=_dev(torch, d["pulse_length"], f64),
        gain_is_table=True, sa_is_table=True, gpt=gpt)


@pytest.mark.parametrize("dtype", ["float64", "float32"])
@pytest.mark.parametrize("cal_type", ["Sv", "TS"])
@pytest.mark.parametrize("shape", [(2, 50, 1000), (3, 17, 333), (1, 4, 4096)])
def test_sv_power_ek60(env, dtype, cal_type, shape):
    torch, ops, synth = env
    d = synth.ek60_numpy(*shape, vary_tau=True)
    d["transmit_duration_nominal"][0, 3] = np.nan  # NaN ping -> NaN row (cal_params.py:290,316)
    exp, exp_r = _oracle_ek60(d, cal_type)
    coef = _coef_ek60(torch, ops, d, cal_type)
    out, rng = ops.sv_power(_dev(torch, d["backscatter_r"]), coef, cal_type=cal_type,
                            dtype=getattr(torch, dtype))
    _assert_close(out.cpu().numpy(), exp, RTOL[dtype], f"{cal_type} {dtype}")
    if dtype == "float64":  # reference operation order (range.py:138) -> bit-identical
        np.testing.assert_array_equal(rng.cpu().numpy(), exp_r)
    else:
        _assert_close(rng.cpu().numpy(), exp_r, 1e-6, "echo_range")
    assert np.isnan(exp[:, :, :3]).all()  # samples 0..2: R' <= 0


@pytest.mark.parametrize("dtype", ["float64", "float32"])
@pytest.mark.parametrize("case", ["uniform_d", "nan_ping", "d_differs"])
@pytest.mark.parametrize("shape", [(2, 50, 1000), (3, 17, 2052), (1, 9, 4096)])
def test_sv_power_one_piece_workgroups_equal_the_strided_rows_kernel(env, monkeypatch, dtype, case, shape):
    """K1's two kernels (round 5): one 1024-sample piece per workgroup (the default) and a workgroup striding over the
    rows of one chunk (EPA_K1_PIECES=0).  Same arithmetic; the cached n log10(s - d) comes from a per-channel table when
    the channel's rows share one d (bit for bit the strided kernel's), else from the table-driven logarithm (1e-13).
    The by-products -- echo_range, its {nanmin, nanmax, NaN count} with and without the array -- are identical."""
    torch, ops, synth = env
    from echopype_amd import _lib

    d = synth.ek60_numpy(*shape, vary_tau=(case == "d_differs"))  # (EK60: d = 2 whatever tau is ...)
    if case == "nan_ping":
        d["transmit_duration_nominal"][0, 3] = np.nan           # a NaN row: its d is NaN, the channel has no table
    coef = _coef_ek60(torch, ops, d, "Sv")
    if case == "d_differs":                                      # (... so the rows' d is edited: two values per channel)
        coef[:, 1::2, _lib.CF_D] += 0.25
    raw = _dev(torch, d["backscatter_r"])
    dt = getattr(torch, dtype)
    outs = {}
    for pieces in ("1", "0"):
        monkeypatch.setenv("EPA_K1_PIECES", pieces)
        with _lib.launch_trace() as tr:
            a = ops.sv_power(raw, coef, dtype=dt)
            b = ops.sv_power(raw, coef, dtype=dt, want_range=False, want_range_stats=True)
            c = ops.sv_power(raw, coef, dtype=dt, want_range=True, want_range_stats=True)
        # (fp64 with the echo_range array or its statistics stays with the strided-rows kernel: the faster one there;
        #  the plain fp64 Sv of the extra call below takes the pieces)
        d0 = ops.sv_power(raw, coef, dtype=dt, want_range=False)[0] if pieces == "1" else None
        assert ("sv_power_piece_kernel" in tr.kernels) == (pieces == "1" and dtype == "float32"), tr.kernels
        assert ("sv_power_kernel" in tr.kernels) == (pieces == "0" or dtype == "float64"), tr.kernels
        if d0 is not None:
            with _lib.launch_trace() as tr2:
                d0 = ops.sv_power(raw, coef, dtype=dt, want_range=False)[0]
            assert tr2.kernels.count("sv_power_piece_kernel") == 1, tr2.kernels
            plain_pieces = d0.cpu().numpy()
        outs[pieces] = [t.cpu().numpy() for t in (a[0], a[1], b[0], b[2], c[0], c[1], c[2])]
    new, old = outs["1"], outs["0"]
    new = [plain_pieces] + new[1:]  # (the Sv of the one-piece kernel, whatever the dtype)
    exact = dtype == "float32" or case == "uniform_d"
    for i in (0, 2, 4):  # Sv
        np.testing.assert_array_equal(np.isnan(new[i]), np.isnan(old[i]))
        if exact:
            np.testing.assert_array_equal(new[i], old[i])
        else:
            f = np.isfinite(old[i])
            assert np.max(np.abs(new[i][f] - old[i][f])) < 1e-11
    for i in (1, 3, 5, 6):  # echo_range, statistics
        np.testing.assert_array_equal(new[i], old[i])
    assert new[3][2] == np.isnan(new[1]).sum()


def test_sv_power_ek80_cw_power_with_gpt(env):
    torch, ops, synth = env
    d = synth.ek60_numpy(2, 30, 512)
    gpt = np.array([True, False])
    tau_eff = np.array([d["transmit_duration_nominal"][0, 0], 0.9e-3])
    gain = ocal.vend_cal_params_power(d["transmit_duration_nominal"], d["pulse_length"], d["gain_correction"])
    sa = ocal.vend_cal_params_power(d["transmit_duration_nominal"], d["pulse_length"], d["sa_correction"])
    exp, exp_r = ocal.cal_power_ek(
        d["backscatter_r"], sonar="EK80", cal_type="Sv", sample_interval=d["sample_interval"],
        sound_speed=d["sound_speed_indicative"], absorption=d["absorption_indicative"],
        transmit_power=d["transmit_power"], tau_nominal=d["transmit_duration_nominal"], gain=gain,
        sa_correction=sa, psi=d["equivalent_beam_angle"], f_nominal=d["frequency_nominal"],
        tau_eff=tau_eff, gpt=gpt)
    f64 = torch.float64
    coef = ops.power_coef_ek(
        _dev(torch, d["sample_interval"]), _dev(torch, d["transmit_duration_nominal"]),
        _dev(torch, d["transmit_power"]), _dev(torch, d["sound_speed_indicative"]),
        _dev(torch, d["absorption_indicative"]), _dev(torch, gain), _dev(torch, sa),
        _dev(torch, d["equivalent_beam_angle"]), _dev(torch, d["frequency_nominal"]), _dev(torch, tau_eff),
        sonar="EK80", cal_type="Sv", gpt=_dev(torch, gpt.astype(np.uint8)))
    out, rng = ops.sv_power(_dev(torch, d["backscatter_r"]), coef, dtype=f64)
    _assert_close(out.cpu().numpy(), exp, 1e-9, "EK80 power Sv")
    np.testing.assert_array_equal(rng.cpu().numpy(), exp_r)


def _time_bins(torch, ops, ping_time, ping_time_bin, closed="left"):
    t_edges = ogrid.ping_edges(ping_time, ping_time_bin)
    ns = t_edges.astype("datetime64[ns]").astype(np.int64)
    n_t = len(ns) - 1
    bs = ops.time_bin_offsets(_dev(torch, ping_time.astype("datetime64[ns]").astype(np.int64)),
                              ns[0], ns[1] - ns[0], n_t, closed=closed)
    return bs, n_t


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_fused_sv_mvbs_ek60(env, dtype):
    torch, ops, synth = env
    d = synth.ek60_numpy(2, 205, 1000)  # 205 pings: last 20-s bin is ragged
    exp_sv, exp_r = _oracle_ek60(d, "Sv")
    exp_mv, _, r_left = ogrid.compute_MVBS(exp_sv, exp_r, d["ping_time"], "1m", "20s")
    coef = _coef_ek60(torch, ops, d, "Sv")
    bs, n_t = _time_bins(torch, ops, d["ping_time"], "20s")
    res = ops.sv_mvbs_fused(_dev(torch, d["backscatter_r"]), coef, bs, n_t, 1.0, len(r_left),
                            dtype=getattr(torch, dtype), want_range=True, want_partials=True)
    _assert_close(res["Sv"].cpu().numpy(), exp_sv, RTOL[dtype], "fused Sv")
    _assert_close(res["echo_range"].cpu().numpy(), exp_r, 1e-12 if dtype == "float64" else 1e-6, "range")
    _assert_close(res["MVBS"].cpu().numpy(), exp_mv, RTOL[dtype], "fused MVBS")
    # partial sums re-finalised == MVBS
    again = ops.mvbs_finalize(res["sum"], res["cnt"])
    _assert_close(again.cpu().numpy(), exp_mv, RTOL[dtype], "finalize(sum,cnt)")


@pytest.mark.parametrize("kind", ["regular", "irregular"])
def test_mvbs_reference_kat(env, kind):
    """The reference's own MVBS fixture (test_commongrid_api.py:363-436): brute-force values
    atol=rtol=1e-10 and NaN mask, through the HIP kernel with a full echo_range array."""
    torch, ops, _ = env
    d = kf.mock_small(kind)
    exp = kf.brute_force_mvbs(d, "1s", 2)
    bs, n_t = _time_bins(torch, ops, d["ping_time"], "1s")
    r_edges = ogrid.range_edges(d["echo_range"], 2.0)
    res = ops.mvbs(_dev(torch, d["Sv"]), bs, n_t, 2.0, len(r_edges) - 1, range=_dev(torch, d["echo_range"]))
    got = res["MVBS"].cpu().numpy()
    assert got.shape == exp.shape
    np.testing.assert_allclose(got, exp, atol=1e-10, rtol=1e-10, equal_nan=True)


@pytest.mark.parametrize("skipna,range_key", [(True, "depth"), (False, "depth"), (True, "echo_range"), (False, "echo_range")])
def test_mvbs_skipna_masks_kat(env, skipna, range_key):
    torch, ops, _ = env
    d = kf.mock_small("irregular")
    sub = {k: (v[:, :2].copy() if v.ndim == 3 else v[:2]) for k, v in d.items()}
    exp, _, r_left = ogrid.compute_MVBS(sub["Sv"], sub[range_key], sub["ping_time"], "2m", "20s", skipna=skipna)
    bs, n_t = _time_bins(torch, ops, sub["ping_time"], "20s")
    res = ops.mvbs(_dev(torch, sub["Sv"]), bs, n_t, 2.0, len(r_left), range=_dev(torch, sub[range_key]),
                   skipna=skipna)
    _assert_close(res["MVBS"].cpu().numpy(), exp, 1e-10, "skipna KAT")


@pytest.mark.parametrize("closed", ["left", "right"])
def test_mvbs_closed_and_edge_membership(env, closed):
    """Samples sitting exactly on float edges (0.2-m bins: arange edge 0.6000000000000001)."""
    torch, ops, _ = env
    rng = np.random.default_rng(5)
    C, P, S = 2, 40, 64
    er = np.tile(np.arange(S) * 0.1, (C, P, 1))  # many samples exactly on k*0.2 edges
    sv = rng.normal(-70, 5, size=(C, P, S))
    pt = kf.gen_ping_time(P, "1s")
    exp, _, r_left = ogrid.compute_MVBS(sv, er, pt, "0.2m", "10s", closed=closed)
    bs, n_t = _time_bins(torch, ops, pt, "10s", closed)
    res = ops.mvbs(_dev(torch, sv), bs, n_t, 0.2, len(r_left), range=_dev(torch, er), closed=closed)
    _assert_close(res["MVBS"].cpu().numpy(), exp, 1e-10, f"closed={closed}")


def test_mvbs_few_bins_two_stage_path(env):
    """One huge ping bin -> pings are split across workgroups and merged through global atomics."""
    torch, ops, _ = env
    d = kf.sv_regular(2, 200, 0.5, 600, "0.3s")
    exp, _, r_left = ogrid.compute_MVBS(d["Sv"], d["echo_range"], d["ping_time"], "5m", "1h")
    bs, n_t = _time_bins(torch, ops, d["ping_time"], "1h")
    assert n_t == 1
    res = ops.mvbs(_dev(torch, d["Sv"]), bs, n_t, 5.0, len(r_left), range=_dev(torch, d["echo_range"]))
    _assert_close(res["MVBS"].cpu().numpy(), exp, 1e-10, "two-stage")


def test_mvbs_range_grid_larger_than_lds(env):
    """20 000 range bins do not fit the LDS budget: accumulation falls back to global atomics."""
    torch, ops, _ = env
    d = kf.sv_regular(2, 2000, 0.5, 120, "1s")
    d["Sv"][0, 7, 100:300] = np.nan
    exp, _, r_left = ogrid.compute_MVBS(d["Sv"], d["echo_range"], d["ping_time"], "0.05m", "30s")
    assert len(r_left) > 16000
    bs, n_t = _time_bins(torch, ops, d["ping_time"], "30s")
    res = ops.mvbs(_dev(torch, d["Sv"]), bs, n_t, 0.05, len(r_left), range=_dev(torch, d["echo_range"]))
    _assert_close(res["MVBS"].cpu().numpy(), exp, 1e-10, "global-atomics path")
    # the fused entry point takes the same fallback (generic kernel) for such a grid
    d2 = env[2].ek60_numpy(1, 40, 2000)
    sv, er = _oracle_ek60(d2, "Sv")
    exp2, _, r_left2 = ogrid.compute_MVBS(sv, er, d2["ping_time"], "0.01m", "20s")
    bs2, n_t2 = _time_bins(torch, ops, d2["ping_time"], "20s")
    res2 = ops.sv_mvbs_fused(_dev(torch, d2["backscatter_r"]), _coef_ek60(torch, ops, d2, "Sv"), bs2, n_t2, 0.01,
                             len(r_left2))
    _assert_close(res2["MVBS"].cpu().numpy(), exp2, 1e-10, "fused, global-atomics path")
    _assert_close(res2["Sv"].cpu().numpy(), sv, 1e-9, "fused Sv")


def test_fused_flag_combinations_take_generic_kernel(env):
    """closed='right' / skipna=False / TS are served by the generic kernel: same answers."""
    torch, ops, synth = env
    d = synth.ek60_numpy(2, 90, 512)
    coef = _coef_ek60(torch, ops, d, "Sv")
    sv, er = _oracle_ek60(d, "Sv")
    for closed, skipna in (("right", True), ("left", False), ("right", False)):
        exp, _, r_left = ogrid.compute_MVBS(sv, er, d["ping_time"], "1m", "20s", closed=closed, skipna=skipna)
        bs, n_t = _time_bins(torch, ops, d["ping_time"], "20s", closed)
        res = ops.sv_mvbs_fused(_dev(torch, d["backscatter_r"]), coef, bs, n_t, 1.0, len(r_left), closed=closed,
                                skipna=skipna)
        _assert_close(res["MVBS"].cpu().numpy(), exp, 1e-9, f"closed={closed} skipna={skipna}")
        _assert_close(res["Sv"].cpu().numpy(), sv, 1e-9, "Sv")


def test_fused_writes_sv_for_pings_outside_every_bin(env):
    """Pings that no time bin covers (grid that starts late / ends early) must still be calibrated."""
    torch, ops, synth = env
    d = synth.ek60_numpy(2, 100, 512)
    coef = _coef_ek60(torch, ops, d, "Sv")
    sv, er = _oracle_ek60(d, "Sv")
    ns = d["ping_time"].astype("datetime64[ns]").astype(np.int64)
    # bins cover only pings 30..69
    bs = ops.time_bin_offsets(_dev(torch, ns), int(ns[30]), 20 * 10**9, 2)
    assert bs.cpu().tolist() == [30, 50, 70]
    for kw in (dict(), dict(closed="right")):  # fast kernel / generic kernel
        if kw:
            bs = ops.time_bin_offsets(_dev(torch, ns), int(ns[30]) - 1, 20 * 10**9, 2, closed="right")
        res = ops.sv_mvbs_fused(_dev(torch, d["backscatter_r"]), coef, bs, 2, 1.0, 64, want_range=bool(kw), **kw)
        _assert_close(res["Sv"].cpu().numpy(), sv, 1e-9, f"Sv outside bins {kw}")
        exp = ogrid.groupby_mean(sv[:, 30:70], er[:, 30:70], d["ping_time"][30:70],
                                 d["ping_time"][30:71:20].astype("datetime64[ns]"), np.arange(0, 65.0, 1.0))
        _assert_close(res["MVBS"].cpu().numpy(), exp, 1e-9, "MVBS of the covered pings")


@pytest.mark.parametrize("S", [333, 1001, 2, 7, 1026])
@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_fused_and_unfused_odd_range_lengths(env, S, dtype):
    """Ragged sizes: S not a multiple of the vector widths (scalar lanes), S smaller than a wavefront,
    S just past one 1024-sample chunk."""
    torch, ops, synth = env
    d = synth.ek60_numpy(2, 47, S)
    sv, er = _oracle_ek60(d, "Sv")
    coef = _coef_ek60(torch, ops, d, "Sv")
    td = getattr(torch, dtype)
    out, rng = ops.sv_power(_dev(torch, d["backscatter_r"]), coef, dtype=td)
    _assert_close(out.cpu().numpy(), sv, RTOL[dtype], f"K1 S={S}")
    if np.isfinite(er).any() and np.nanmax(er) > 0:
        exp_mv, _, r_left = ogrid.compute_MVBS(sv, er, d["ping_time"], "0.5m", "20s")
        bs, n_t = _time_bins(torch, ops, d["ping_time"], "20s")
        res = ops.sv_mvbs_fused(_dev(torch, d["backscatter_r"]), coef, bs, n_t, 0.5, len(r_left), dtype=td)
        _assert_close(res["Sv"].cpu().numpy(), sv, RTOL[dtype], f"fused Sv S={S}")
        _assert_close(res["MVBS"].cpu().numpy(), exp_mv, RTOL[dtype], f"fused MVBS S={S}")


def test_empty_and_degenerate_inputs(env):
    """All-NaN pings, a dataset whose every sample is NaN, a single ping / single sample."""
    torch, ops, synth = env
    d = synth.ek60_numpy(1, 40, 64)
    d["backscatter_r"][:, 10:20, :] = np.nan
    sv, er = _oracle_ek60(d, "Sv")
    coef = _coef_ek60(torch, ops, d, "Sv")
    exp_mv, _, r_left = ogrid.compute_MVBS(sv, er, d["ping_time"], "1m", "10s")
    bs, n_t = _time_bins(torch, ops, d["ping_time"], "10s")
    res = ops.sv_mvbs_fused(_dev(torch, d["backscatter_r"]), coef, bs, n_t, 1.0, len(r_left))
    _assert_close(res["MVBS"].cpu().numpy(), exp_mv, 1e-9, "all-NaN time bin")
    assert np.isnan(res["MVBS"].cpu().numpy()[0, 1]).all()  # pings 10..19 = one empty bin -> fill_value
    res = ops.sv_mvbs_fused(_dev(torch, d["backscatter_r"]), coef, bs, n_t, 1.0, len(r_left), fill_value=-999.0)
    assert (res["MVBS"].cpu().numpy()[0, 1] == -999.0).all()
    allnan = np.full((1, 5, 8), np.nan, dtype=np.float32)
    d1 = synth.ek60_numpy(1, 5, 8)
    out, rng = ops.sv_power(_dev(torch, allnan), _coef_ek60(torch, ops, d1, "Sv"))
    assert np.isnan(out.cpu().numpy()).all() and np.isnan(rng.cpu().numpy()).all()
    lo, hi = ops.nanminmax(rng)
    assert np.isnan(lo) and np.isnan(hi)
    d2 = synth.ek60_numpy(1, 1, 4)
    sv2, _ = _oracle_ek60(d2, "Sv")
    out2, _ = ops.sv_power(_dev(torch, d2["backscatter_r"]), _coef_ek60(torch, ops, d2, "Sv"))
    _assert_close(out2.cpu().numpy(), sv2, 1e-9, "1 ping x 4 samples")


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_fused_int16_ingest_is_bit_identical_to_the_f32_path(env, dtype):
    """feeding the instrument's int16 samples + recorded ping lengths gives exactly
    what the converter's float32 NaN-padded array gives (parse_base.py:24,302 arithmetic in-kernel)."""
    torch, ops, synth = env
    rng = np.random.default_rng(9)
    C, P, S = 2, 260, 1024
    d = synth.ek60_numpy(C, P, S)
    i16 = rng.integers(-12000, -2000, size=(C, P, S), dtype=np.int16)
    n_valid = np.full((C, P), S, dtype=np.int32)
    n_valid[:, rng.random(P) < 0.2] = 970
    n_valid[1, 7] = 0          # an empty ping
    n_valid[0, 9] = 333        # odd length: the boundary falls inside a sample pair
    f32 = i16.astype("float32") * synth.INDEX2POWER      # the converter's arithmetic
    f32[np.arange(S)[None, None, :] >= n_valid[:, :, None]] = np.nan
    d["backscatter_r"] = f32
    coef = _coef_ek60(torch, ops, d, "Sv")
    bs, n_t = _time_bins(torch, ops, d["ping_time"], "20s")
    td = getattr(torch, dtype)
    a = ops.sv_mvbs_fused(_dev(torch, f32), coef, bs, n_t, 1.0, 200, dtype=td, want_partials=True)
    b = ops.sv_mvbs_fused_i16(_dev(torch, i16), _dev(torch, n_valid), coef, bs, n_t, 1.0, 200, dtype=td,
                              want_partials=True, want_range_max=True)
    np.testing.assert_array_equal(b["Sv"].cpu().numpy(), a["Sv"].cpu().numpy())
    np.testing.assert_array_equal(b["cnt"].cpu().numpy(), a["cnt"].cpu().numpy())
    _assert_close(b["MVBS"].cpu().numpy(), a["MVBS"].cpu().numpy(), 1e-12 if dtype == "float64" else 1e-5, "MVBS")
    sv, er = _oracle_ek60(d, "Sv")
    _assert_close(b["Sv"].cpu().numpy(), sv, RTOL[dtype], "int16 ingest vs oracle")
    assert float(b["range_max"].item()) == np.nanmax(er)


def test_mvbs_index_binning_kat(env):
    # test_commongrid_api.py:171-202 shape (4,100,4000) with ping_num=3, range_sample_num=7
    torch, ops, _ = env
    d = kf.sv_regular(4, 4000, 0.5, 100)
    d["Sv"][1, 5, 100:140] = np.nan
    exp, exp_r = ogrid.compute_MVBS_index_binning(d["Sv"], d["echo_range"], 7, 3)
    got, rmin = ops.mvbs_index(_dev(torch, d["Sv"]), 3, 7, range=_dev(torch, d["echo_range"]))
    _assert_close(got.cpu().numpy(), exp, 1e-10, "index binning")
    _assert_close(rmin.cpu().numpy(), exp_r, 1e-15, "echo_range block min")


def test_noise_reference_kat(env):
    torch, ops, _ = env
    for make, nan_expect in ((kf.noise_toy, None), (kf.noise_seed1, 6)):
        Sv, er, a = make()
        exp_n, exp_c = oclean.remove_background_noise(Sv, er, a, 2, 5, SNR_threshold="0dB")
        a2 = _dev(torch, np.full((1, 10), 2 * a))
        nb = ops.noise_estimate(_dev(torch, Sv), a2, 2, 5, range=_dev(torch, er))
        sn, sc = ops.noise_apply(_dev(torch, Sv), a2, nb, 2, 0.0, range=_dev(torch, er))
        _assert_close(sn.cpu().numpy(), exp_n, 1e-10, "Sv_noise")
        _assert_close(sc.cpu().numpy(), exp_c, 1e-9, "Sv_corrected")
        c = sc.cpu().numpy()
        if nan_expect is None:
            assert np.isnan(c[0, 0, 30]) and np.isnan(c[0, 0, 60])  # test_noise.py:943-948
        else:
            assert np.count_nonzero(np.isnan(c[0, :, :50])) == nan_expect  # test_noise.py:983-987


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_noise_on_calibrated_ek60(env, dtype):
    torch, ops, synth = env
    d = synth.ek60_numpy(2, 101, 1000)
    sv, er = _oracle_ek60(d, "Sv")
    alpha = d["absorption_indicative"]
    exp_n, exp_c = oclean.remove_background_noise(sv, er, alpha, 20, 50, background_noise_max="-125dB",
                                                  SNR_threshold="3.0dB")
    td = getattr(torch, dtype)
    svd, erd = _dev(torch, sv, td), _dev(torch, er, td)
    a2 = _dev(torch, 2 * alpha)
    nb = ops.noise_estimate(svd, a2, 20, 50, range=erd, noise_max=-125.0)
    sn, sc = ops.noise_apply(svd, a2, nb, 20, 3.0, range=erd)
    if dtype == "float64":
        _assert_close(sn.cpu().numpy(), exp_n, 1e-9, "Sv_noise")
        _assert_close(sc.cpu().numpy(), exp_c, 1e-7, "Sv_corrected")
    else:
        # fp32: thresholded NaN pattern may flip for samples within rounding of the SNR threshold
        g, e = sc.cpu().numpy().astype(np.float64), exp_c
        both = ~np.isnan(g) & ~np.isnan(e)
        assert (np.isnan(g) != np.isnan(e)).mean() < 1e-3
        assert np.max(np.abs(g[both] - e[both]) / np.abs(e[both])) < 1e-3
        _assert_close(sn.cpu().numpy(), exp_n, 1e-3, "Sv_noise f32")


def test_fast_exp10_accuracy(env):
    """The fused kernel's table-driven 10^(u/10) vs numpy's in extended precision: <= 2 ulp."""
    torch, ops, _ = env
    rng = np.random.default_rng(3)
    u = np.concatenate([rng.uniform(-200, 60, 200000), rng.uniform(-3200, 3200, 20000),
                        np.array([0.0, -0.0, 10.0, -10.0, 1e-300, np.nan, np.inf, -np.inf, 3100.0, -3300.0])])
    got = ops.selftest_lin_from_db(_dev(torch, u)).cpu().numpy()
    exp = np.power(np.longdouble(10.0), np.longdouble(u) / np.longdouble(10.0))
    fin = np.isfinite(u) & (np.abs(u) < 3000)
    rel = np.abs((np.longdouble(got[fin]) - exp[fin]) / exp[fin]).astype(np.float64)
    assert rel.max() < 4.5e-16, rel.max()
    assert np.isnan(got[np.isnan(u)]).all()
    assert got[u == np.inf][0] == np.inf and got[u == -np.inf][0] == 0.0
    assert got[u == 3100.0][0] == np.inf and got[u == -3300.0][0] == 0.0


@pytest.mark.parametrize("dtype", ["float64", "float32"])
@pytest.mark.parametrize("S", [1000, 1001, 37])
def test_sv_power_range_stats_by_product(env, dtype, S):
    """epa_sv_power_stats: {nanmin, nanmax, NaN count} of the echo_range it writes == epa_nanminmax of that array
    (vector path, and the scalar path of odd sizes); outputs identical to epa_sv_power."""
    torch, ops, synth = env
    d = synth.ek60_numpy(2, 33, S, seed=S)
    coef = _coef_ek60(torch, ops, d, "Sv")
    raw = _dev(torch, d["backscatter_r"])
    dt = getattr(torch, dtype)
    sv0, rg0 = ops.sv_power(raw, coef, dtype=dt)
    sv1, rg1, st = ops.sv_power(raw, coef, dtype=dt, want_range_stats=True)
    assert torch.equal(torch.nan_to_num(sv0, nan=1.0), torch.nan_to_num(sv1, nan=1.0))
    assert torch.equal(torch.nan_to_num(rg0, nan=-1.0), torch.nan_to_num(rg1, nan=-1.0))
    lo, hi, nn = ops.nanminmax(rg0, with_nan_count=True)
    assert st.cpu().tolist() == [lo, hi, float(nn)]
    assert nn > 0 and hi > lo


@pytest.mark.parametrize("inline", [False, True])
def test_fast_log10_accuracy(env, inline):
    """Table-driven f64 log10 (and its call-free variant) vs an extended-precision reference: absolute error
    <= 2e-16 * max(1, |log10 x|), specials exact."""
    torch, ops, _ = env
    rng = np.random.default_rng(4)
    x = np.concatenate([10 ** rng.uniform(-30, 30, 200000), rng.uniform(0.5, 2.0, 100000),
                        1 + rng.uniform(-1e-6, 1e-6, 1000), 2.0 ** np.arange(-1000, 1000, 37.0),
                        np.array([1.0, 2.0, 0.5, np.sqrt(2), 10.0, 5e-324, 1e-310, 0.0, -1.0, np.inf, np.nan])])
    got = ops.selftest_log10(_dev(torch, x), inline=inline).cpu().numpy()
    with np.errstate(all="ignore"):
        exp = np.log10(np.longdouble(x)).astype(np.longdouble)
    fin = np.isfinite(x) & (x > 0)
    err = np.abs(np.longdouble(got[fin]) - exp[fin]).astype(np.float64)
    bound = 2.3e-16 * np.maximum(1.0, np.abs(exp[fin].astype(np.float64)))
    assert (err <= bound).all(), float((err / bound).max())
    assert got[x == 1.0][0] == 0.0 and abs(got[x == 10.0][0] - 1.0) < 3e-16
    assert got[x == 0.0][0] == -np.inf and np.isnan(got[x == -1.0][0]) and got[x == np.inf][0] == np.inf
    assert np.isnan(got[np.isnan(x)]).all()


def test_argument_errors_from_c_abi(env):
    torch, ops, synth = env
    d = synth.ek60_numpy(1, 4, 64)
    coef = _coef_ek60(torch, ops, d, "Sv")
    with pytest.raises(ValueError, match="float32"):
        ops.sv_power(_dev(torch, d["backscatter_r"]).double(), coef)
    with pytest.raises(ValueError, match="device"):
        ops.sv_power(torch.from_numpy(d["backscatter_r"]), coef)
    from echopype_amd import _lib
    with pytest.raises(ValueError, match="NULL"):
        _lib.call("epa_sv_power", None, None, 1, 1, 1, 0, 0, None, None, 1, None)


# ---- EK80 BB: FFT path == direct path ---------------------------------------------------------------
@pytest.mark.parametrize("in_dtype,out_dtype,fft_dtype", [("float64", "float64", None), ("float32", "float64", None),
                                                          ("float32", "float32", None), ("float32", "float32", "float64"),
                                                          ("float32", "float64", "float32"), ("float64", "float32", None)])
@pytest.mark.parametrize("taps,S,mixed,B", [(177, 5000, False, 4), (64, 2048, True, 4), (16, 1873, False, 4),
                                            (1024, 3000, True, 4), (333, 8192, False, 4), (90, 2500, True, 3),
                                            (40, 1000, False, 1), (177, 2100, True, 4), (100, 1949, False, 4),
                                            (31, 300, True, 2)])
def test_sv_complex_fft_path_matches_direct(env, in_dtype, out_dtype, fft_dtype, taps, S, mixed, B):
    """The LDS-FFT circular correlation (epa_sv_complex_fft) against the sliding-window direct form
    (epa_sv_complex) on echoes spanning 140 dB: several tiles, per-sector second pass
    for mixed NaN patterns, NaN tails, two channels with different replica lengths; complex64 and complex128
    transforms (default: the output's precision)."""
    torch, ops, synth = env
    rng = np.random.default_rng(taps + S)
    C, P = 2, 5
    amp = 10.0 ** rng.uniform(-7, 0, (C, P, S, 1))
    re = (amp * rng.standard_normal((C, P, S, B))).astype(in_dtype)
    im = (amp * rng.standard_normal((C, P, S, B))).astype(in_dtype)
    re[:, :, S - 37:], im[:, :, S - 37:] = np.nan, np.nan  # end-of-ping padding
    re[1, 1], im[1, 1] = np.nan, np.nan                    # a whole missing ping
    if mixed:
        re[0, 0, 100:130, B - 1] = np.nan                  # one sector missing -> per-sector fallback
        im[0, 2, S // 2, 0] = np.nan
        re[1, 0, 200:203, 0] = np.nan                      # beam 0 missing: masked echo_range, Sv NaN (B > 1: others valid)
    lens = [taps, max(taps // 2, 1)]
    rep = np.concatenate([(rng.standard_normal(n) + 1j * rng.standard_normal(n)) * np.hanning(n + 2)[1:-1]
                          for n in lens]).astype(np.complex64)
    repf = _dev(torch, np.stack([rep.real, rep.imag], axis=1).astype(np.float32).reshape(-1))
    off = _dev(torch, np.array([0, lens[0], lens[0] + lens[1]], dtype=np.int32))
    cc = np.zeros((C, P, 8))
    cc[..., 0], cc[..., 1], cc[..., 2], cc[..., 3], cc[..., 4], cc[..., 5] = 2.6e-5, 750.0, 0.2, 0.02, -30.0, 1e3
    # a ping with its own sound speed / absorption: it leaves the per-channel time-varied-gain table (and its
    # neighbours in a tile do not)
    cc[0, 2, 1], cc[1, 2, 3] = 751.5, 0.021
    cc[:, 3, 4], cc[:, 4, 5] = -31.5, 1.1e3                 # per-ping gain / power terms
    kw = dict(replica=repf, replica_off=off, max_taps=taps, dtype=getattr(torch, out_dtype), want_prx=True)
    args = (_dev(torch, re), _dev(torch, im), _dev(torch, cc))
    d = ops.sv_complex(*args, method="direct", **kw)
    f = ops.sv_complex(*args, method="fft", fft_dtype=fft_dtype, want_range_stats=True, **kw)
    er = f["echo_range"].cpu().numpy().astype(np.float64)
    st = f["range_stats"].cpu().numpy()
    assert st[2] == np.isnan(er).sum() and st[0] == np.nanmin(er) and st[1] == np.nanmax(er)
    pd, pf = d["prx"].cpu().numpy().astype(np.float64), f["prx"].cpu().numpy().astype(np.float64)
    np.testing.assert_array_equal(f["echo_range"].cpu().numpy(), d["echo_range"].cpu().numpy())
    with np.errstate(invalid="ignore"), warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        peak = np.nanmax(pd, axis=2, keepdims=True)  # (every ping is tiled on its own)
    fft32 = (fft_dtype or out_dtype) == "float32"
    differ = np.isnan(pf) != np.isnan(pd)
    if fft32:
        # 130 dB under the ping's strongest echo a complex64 transform returns rounding noise, which may be an exact 0;
        # the reference turns a received power that is not > 0 into NaN (calibrate_ek.py:581 prx.where(prx > 0, nan)),
        # and so do both kernels.  The only NaN difference allowed is that one: NaN from the complex64 transform where
        # the direct form (accumulating in the output precision) still holds a positive value that small -- never the
        # reverse, never on a sample of any strength.
        with np.errstate(invalid="ignore"):
            differ &= ~(np.isnan(pf) & (pd < 1e-13 * peak))
    assert not differ.any(), np.argwhere(differ)[:5]
    # the direct path accumulates in the output precision; the transform in fft_dtype (default: the same).
    # Error model: the amplitude error of a transform is delta = eps_a * (strongest echo of the ping) whatever the
    # sample, so |d prx| <= delta * (2 sqrt(prx) + delta) (+ the relative rounding of the epilogue)
    f32 = out_dtype == "float32" or fft_dtype == "float32"
    eps_a = 1e-12 if not f32 else 2e-6
    with np.errstate(invalid="ignore"):
        delta = eps_a * np.sqrt(peak)
        bound = delta * (2 * np.sqrt(pd) + delta) + eps_a * pd + 1e-300
        assert np.nanmax(np.abs(pf - pd) / bound) < 1.0
    sd, sf = d["out"].cpu().numpy().astype(np.float64), f["out"].cpu().numpy().astype(np.float64)
    assert not ((np.isnan(sf) != np.isnan(sd)) & ~(np.isnan(pf) != np.isnan(pd))).any()  # (Sv is NaN where prx is)
    # dB values agree wherever the sample is within 100 dB (complex128 transform) / 40 dB (complex64 transform:
    # its error is relative to the tile's strongest echo) of the ping's strongest echo
    with np.errstate(invalid="ignore"):
        strong = pd > peak * (1e-4 if fft32 else 1e-10)
    tol = 1e-6 if not f32 else 2e-3
    assert np.nanmax(np.abs(sf[strong] - sd[strong])) < tol  # NaN where R' <= 0 (both paths alike)


@pytest.mark.parametrize("in_dtype", ["float32", "float64"])
@pytest.mark.parametrize("out_dtype", ["float64", "float32"])
@pytest.mark.parametrize("B,S,offset", [(4, 5000, 0), (4, 2049, 1), (3, 700, 0), (1, 300, 0), (6, 1000, 0)])
def test_sv_complex_cw_streaming_kernel(env, in_dtype, out_dtype, B, S, offset):
    """epa_sv_complex without a replica (CW: the streaming kernel) against a NumPy statement of calibrate_ek.py:483-490,
    571-638: NaN-skipping sector mean, prx > 0 else NaN, masked echo_range (beam-0 real part), R' <= 0 -> NaN.  Four
    sectors on 16-byte aligned planes take vector loads, every other sector count / an unaligned plane the scalar
    form (offset = 1 element shifts the planes off the 16-byte boundary); ragged last piece, NaN tails, partly-NaN
    samples, a missing beam 0."""
    torch, ops, synth = env
    rng = np.random.default_rng(B * 1000 + S)
    C, P = 2, 5
    amp = 10.0 ** rng.uniform(-6, 0, (C, P, S, 1))
    re = (amp * rng.standard_normal((C, P, S, B))).astype(in_dtype)
    im = (amp * rng.standard_normal((C, P, S, B))).astype(in_dtype)
    re[:, 1, S - 40:], im[:, 1, S - 40:] = np.nan, np.nan
    re[1, 2], im[1, 2] = np.nan, np.nan
    re[0, 0, 50:60, B - 1] = np.nan       # one sector missing (the only one when B == 1)
    im[0, 3, 100, 0] = np.nan             # imaginary part of beam 0 only: the range stays valid
    re[1, 4, 200:203, 0] = np.nan         # beam 0 missing: echo_range and Sv NaN
    cc = np.zeros((C, P, 8))
    cc[..., 0], cc[..., 1], cc[..., 2], cc[..., 3], cc[..., 4], cc[..., 5] = 2.6e-5, 750.0, 0.2, 0.02, -30.0, 1e3
    cc[..., 1] += rng.uniform(-1, 1, (C, P))
    def dev_plane(a):  # optionally off the 16-byte boundary
        flat = torch.empty(a.size + offset, dtype=getattr(torch, in_dtype), device="cuda")
        flat[offset:] = torch.from_numpy(a.reshape(-1)).cuda()
        return flat[offset:].view(a.shape)
    res = ops.sv_complex(dev_plane(re), dev_plane(im), _dev(torch, cc), dtype=getattr(torch, out_dtype), want_prx=True)
    # ---- the reference's arithmetic
    z = re.astype(np.float64) + 1j * im.astype(np.float64)
    with np.errstate(invalid="ignore"), warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        mean_b = np.nanmean(np.where(np.isnan(z), np.nan, z), axis=3)
        prx = cc[..., 5:6] * np.abs(mean_b) ** 2
        prx = np.where(prx > 0, prx, np.nan)
        R = (np.arange(S)[None, None, :] * cc[..., 0:1]) * cc[..., 1:2]
        R = np.where(np.isnan(re[..., 0]), np.nan, R)
        Rt = R - cc[..., 2:3]
        Rt = np.where(Rt > 0, Rt, np.nan)
        exp = 10 * np.log10(prx) + 20 * np.log10(Rt) + cc[..., 3:4] * Rt + cc[..., 4:5]
    got = res["out"].cpu().numpy().astype(np.float64)
    np.testing.assert_array_equal(np.isnan(got), np.isnan(exp))
    ok = ~np.isnan(exp)
    f32 = "float32" in (in_dtype, out_dtype) and out_dtype == "float32"
    np.testing.assert_allclose(got[ok], exp[ok], rtol=0, atol=2e-3 if f32 else 1e-9)
    er = res["echo_range"].cpu().numpy().astype(np.float64)
    np.testing.assert_array_equal(np.isnan(er), np.isnan(R))
    np.testing.assert_allclose(er[~np.isnan(R)], R[~np.isnan(R)], rtol=1e-6 if out_dtype == "float32" else 0, atol=0)
    assert ok.mean() > 0.5


# ---- the whole chain in two passes ------------------------------------------------------------------
@pytest.mark.parametrize("dtype", ["float64", "float32"])
@pytest.mark.parametrize("closed,few_bins,pn,bin_s", [("left", False, 20, 20), ("right", False, 20, 20),
                                                       ("left", True, 20, 20), ("left", False, 80, 100)])
@pytest.mark.parametrize("ss_every", [1, 50])
def test_fused_chain_equals_the_four_separate_kernels(env, dtype, closed, few_bins, pn, bin_s, ss_every):
    """epa_sv_noise_fused == epa_sv_power + epa_noise_estimate and epa_denoise_mvbs == epa_noise_apply +
    epa_mvbs (same arithmetic, fewer sweeps), and both equal the oracle chain.  The (80, 100) case has more
    pings per noise block / time bin than the 64 per-ping logs a workgroup caches.  ss_every = 1: a new sound speed
    with every ping (no two pings share a range vector); 50: most time bins hold pings of one range vector (the
    fp64 pass 2 takes them with per-column constants, chain_fast.hip: sv_denoise_mvbs_uniform_kernel), the bins that
    straddle a change go to the general kernel in the same call."""
    _chain_equivalence(env, dtype, closed, 45 if few_bins else 203, pn, bin_s, ss_every)


@pytest.mark.parametrize("dtype", ["float64", "float32"])
@pytest.mark.parametrize("bin_s", [100, 300])
def test_fused_chain_long_uniform_groups(env, bin_s, dtype):
    """Time bins of 100 pings (beyond the 64 per-ping logs of the general kernel, inside the 256 per-ping constants of
    the uniform-group kernel) and of 300 pings (beyond both: the uniform-group kernel hands the bin over) on a file
    whose pings share one range vector."""
    _chain_equivalence(env, dtype, "left", 620, 20, bin_s, 100000, S=512)


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_fused_chain_with_a_fine_range_grid(env, dtype):
    """0.1-m range bins over 190 m: 1900 bins -- one row of LDS accumulators per workgroup instead of the two the
    uniform-bin kernel otherwise keeps."""
    _chain_equivalence(env, dtype, "left", 83, 20, 20, 100000, S=1000, rbin=0.1)


@pytest.mark.parametrize("dtype", ["float64", "float32"])
@pytest.mark.parametrize("S", [260, 2052, 1000])
@pytest.mark.parametrize("rbin", [1.0, 0.07])
def test_fused_chain_sound_speed_jitter_in_a_partial_last_wavefront(env, S, rbin, dtype):
    """The sound-speed-drift pass 2 (chain_fast.hip: sv_denoise_mvbs_drift_kernel) redoes the columns whose range
    crosses a range-bin edge inside the time bin with the wavefront's lanes spread over the PINGS.  S = 260 / 2052 leave
    two lanes in the row's last wavefront: the pings must be shared among the lanes that exist (round-3 ADVICE: a
    stride of 64 dropped the pings of the missing lanes).  8 m/s of jitter moves the far columns by metres."""
    # time bins of 7 pings: the planner keeps bins of up to 8 pings whole (block_reduce.hip make_plan), which is what the
    # specialised pass-2 kernels serve; the launch trace asserts they ran
    _chain_equivalence(env, dtype, "left", 143, 20, 7, 1, S=S, rbin=rbin, ss_jitter=8.0,
                       expect_kernels=("sv_denoise_mvbs_uniform_kernel", "sv_denoise_mvbs_drift_kernel"))


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_fused_chain_with_empty_time_bins(env, dtype):
    """A 130-s hole in the pings: time bins without a single ping between bins of uniform pings (the uniform-group
    pass 2 writes their fill value) -- and an all-NaN ping block in the noise estimate."""
    _chain_equivalence(env, dtype, "left", 140, 20, 20, 100000, S=512, gap_after=60)


def _chain_equivalence(env, dtype, closed, P, pn, bin_s, ss_every, S=1000, gap_after=None, rbin=1.0, ss_jitter=0.0,
                       expect_kernels=()):
    from echopype_amd import _lib
    torch, ops, synth = env
    C = 2
    d = synth.ek60_numpy(C, P, S, ss_every=ss_every)
    if ss_jitter:  # a recorded sound speed that moves by metres per second from ping to ping: a column's range then
        # crosses range-bin edges inside a time bin (the columns the drift kernel redoes with its lanes over the pings)
        d["sound_speed_indicative"] = d["sound_speed_indicative"] + ss_jitter * np.random.default_rng(5).random((1, P))
    if gap_after is not None:
        d["ping_time"] = d["ping_time"].copy()
        d["ping_time"][gap_after:] += np.timedelta64(130, "s")
        d["backscatter_r"][:, 20:40, :] = np.nan
    d["transmit_power"] = d["transmit_power"] * (1.0 + 0.1 * (np.arange(P) % 7 == 3))  # a per-ping term that may vary
    dt = getattr(torch, dtype)
    g = lambda k: _dev(torch, d[k], torch.float64)  # noqa: E731
    coef = ops.power_coef_ek(g("sample_interval"), g("transmit_duration_nominal"), g("transmit_power"),
                             g("sound_speed_indicative"), g("absorption_indicative"), g("gain_correction"),
                             g("sa_correction"), g("equivalent_beam_angle"), g("frequency_nominal"),
                             _dev(torch, d["transmit_duration_nominal"][:, 0].copy(), torch.float64),
                             pulse_length=g("pulse_length"), gain_is_table=True, sa_is_table=True)
    raw = _dev(torch, d["backscatter_r"])
    a2 = _dev(torch, np.broadcast_to(2 * d["absorption_indicative"], (C, P)).copy(), torch.float64)
    # reference kernels
    sv0, rg0 = ops.sv_power(raw, coef, dtype=dt)
    n0 = ops.noise_estimate(sv0, a2, pn, 50, range=rg0, noise_max=-120.0)
    sn0, sc0 = ops.noise_apply(sv0, a2, n0, pn, 3.0, range=rg0)
    ns = d["ping_time"].astype("datetime64[ns]").astype(np.int64)
    dt_ns = bin_s * 1_000_000_000
    n_t = int((ns[-1] - ns[0]) // dt_ns) + 1
    bs = ops.time_bin_offsets(_dev(torch, ns), int(ns[0]), dt_ns, n_t, closed=closed)
    _, rmax = ops.nanminmax(rg0)
    n_r = len(np.arange(0, rmax + rbin, rbin)) - 1
    m0 = ops.mvbs(sc0, bs, n_t, rbin, n_r, range=rg0, closed=closed)["MVBS"]
    # fp32 stores echo_range rounded to float32, the coefficient rows carry it in double: samples next to a
    # bin edge may change bins, so the affine / raw variants are held to the affine-binned reference
    m0c = ops.mvbs(sc0, bs, n_t, rbin, n_r, coef=coef, closed=closed)["MVBS"]
    # fused pair (transmission loss / bins from the coefficient rows: no echo_range array is read)
    sv1, rg1, n1 = ops.sv_noise_fused(raw, coef, a2, pn, 50, dtype=dt, noise_max=-120.0, want_range=True)
    assert torch.equal(torch.nan_to_num(sv1, nan=1.0), torch.nan_to_num(sv0, nan=1.0))
    assert torch.equal(torch.nan_to_num(rg1, nan=-1.0), torch.nan_to_num(rg0, nan=-1.0))
    _assert_close(n1.cpu().numpy(), n0.cpu().numpy(), 1e-12 if dtype == "float64" else 1e-5, "noise estimate")
    tol = 1e-11 if dtype == "float64" else 2e-4
    variants = [("range", lambda: ops.denoise_mvbs(sv1, a2, n1, pn, 3.0, bs, n_t, rbin, n_r, closed=closed,
                                                    want_noise=True, range=rg0)),
                ("coef", lambda: ops.denoise_mvbs(sv1, a2, n1, pn, 3.0, bs, n_t, rbin, n_r, closed=closed,
                                                   want_noise=True, coef=coef)),
                ("raw", lambda: ops.sv_denoise_mvbs(raw, coef, a2, n1, pn, 3.0, bs, n_t, rbin, n_r, closed=closed,
                                                    dtype=dt, want_noise=True, want_range=True)),
                # without echo_range out: the specialised two-pass kernels (csrc/chain_fast.hip) when closed="left"
                ("raw-fast", lambda: ops.sv_denoise_mvbs(raw, coef, a2, ops.sv_noise_fused(
                    raw, coef, a2, pn, 50, dtype=dt, noise_max=-120.0)[2], pn, 3.0, bs, n_t, rbin, n_r, closed=closed,
                    dtype=dt, want_noise=True))]
    for name, run in variants:
        with _lib.launch_trace() as tr:
            res = run()
        if name == "raw-fast":
            assert all(k in tr.kernels for k in expect_kernels), tr.kernels
        got_n, exp_n0 = res["Sv_noise"].cpu().numpy(), sn0.cpu().numpy()
        if name == "coef":  # affine range: finite where the masked echo_range (and the reference's Sv_noise) is NaN
            got_n = np.where(np.isnan(exp_n0), np.nan, got_n)
        _assert_close(got_n, exp_n0, tol, f"{name}: Sv_noise")
        _assert_close(res["Sv_corrected"].cpu().numpy(), sc0.cpu().numpy(), RTOL[dtype], f"{name}: Sv_corrected")
        _assert_close(res["MVBS"].cpu().numpy(), (m0 if name == "range" else m0c).cpu().numpy(), RTOL[dtype],
                      f"{name}: MVBS of Sv_corrected")
        if name == "raw":
            assert torch.equal(torch.nan_to_num(res["echo_range"], nan=-1.0), torch.nan_to_num(rg0, nan=-1.0))
    if dtype == "float64" and closed == "left":
        sv, er = _oracle_ek60(d, "Sv")
        exp_n, exp_c = oclean.remove_background_noise(sv, er, d["absorption_indicative"], pn, 50, "-120.0dB", "3.0dB")
        exp_m, _, _ = ogrid.compute_MVBS(exp_c, er, d["ping_time"], f"{rbin}m", f"{bin_s}s")
        _assert_close(res["Sv_corrected"].cpu().numpy(), exp_c, 1e-9, "oracle Sv_corrected")
        _assert_close(res["MVBS"].cpu().numpy(), exp_m, 1e-9, "oracle MVBS")


# ---- EK80 complex kernels against the reference's own _cal_complex_samples outputs ---------------------
@pytest.mark.parametrize("tag,wf,planes", [("ek80bb", "BB", "float64"), ("ek80cw", "CW", "float64"),
                                           ("ek80bbseam", "BB", "float32"), ("ek80bbseam", "BB", "float64")])
@pytest.mark.parametrize("method", ["direct", "fft"])
def test_sv_complex_matches_reference_method_goldens(env, tag, wf, planes, method):
    """epa_sv_complex / epa_sv_complex_fft on the inputs of tests/golden/ref_chain_goldens.npz vs the Sv / TS the
    reference's CalibrateEK80._cal_complex_samples produced for them (oracle/gen_chain_goldens.py).  ``ek80bbseam``
    (tests/golden/ref_seam_goldens.npz): 4200 samples per ping -- echoes straddling the overlap-save tile seams of the
    LDS-FFT kernel (samples 1872, 2048, 3744, 4096 +- taps), partly-NaN sectors and a NaN beam 0 inside the overlap
    regions; its inputs are float32 numbers, fed as float32 planes and as the converter's float64 planes."""
    import os

    torch, ops, synth = env
    from echopype_amd import _lib

    gdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    g = np.load(os.path.join(gdir, "ref_seam_goldens.npz" if tag.endswith("seam") else "ref_chain_goldens.npz"))
    C, P, S, B = g[f"{tag}_re"].shape
    reps = [g[f"{tag}_replica{i}"] for i in range(C)]
    cw, si, pt = g[f"{tag}_sound_speed"], g[f"{tag}_sample_interval"], g[f"{tag}_transmit_power"]
    tau = np.tile(g[f"{tag}_tau"][:, None], (1, P))
    lam = cw / g[f"{tag}_f_center"][:, None]
    gain = g[f"{tag}_gain"]
    if wf == "BB":
        gain = gain - ocal.b_theta_phi_m(g[f"{tag}_angle_offset_alongship"], g[f"{tag}_angle_offset_athwartship"],
                                         g[f"{tag}_beamwidth_alongship"], g[f"{tag}_beamwidth_athwartship"])[:, None]
    z_er, z_et = 5400.0, 75.0
    for cal in ("Sv", "TS"):
        cc = np.zeros((C, P, _lib.NCCOEF))
        cc[..., _lib.CC_RA], cc[..., _lib.CC_RB] = si, cw / 2
        cc[..., _lib.CC_SHIFT], cc[..., _lib.CC_ALPHA2] = cw * tau / 4, 2 * g[f"{tag}_absorption"]
        cc[..., _lib.CC_PSCALE] = B / 8.0 * (abs(z_er + z_et) / z_er) ** 2 / z_et
        if cal == "Sv":
            A = (-10 * np.log10(lam**2 * pt * cw / (32 * np.pi**2)) - 2 * gain
                 - 10 * np.log10(g[f"{tag}_tau_effective"])[:, None] - g[f"{tag}_psi"][:, None])
            if wf == "CW":
                A = A - 2 * g[f"{tag}_sa"]
        else:
            A = -10 * np.log10(lam**2 * pt / (16 * np.pi**2)) - 2 * gain
        cc[..., _lib.CC_A] = A
        kw = {}
        if wf == "BB":
            rep = np.concatenate(reps).astype(np.complex64)
            kw = dict(replica=_dev(torch, np.stack([rep.real, rep.imag], axis=1).astype(np.float32).reshape(-1)),
                      replica_off=_dev(torch, np.cumsum([0] + [r.size for r in reps]).astype(np.int32)),
                      max_taps=max(r.size for r in reps), method=method)
        elif method == "fft":
            continue  # CW has no replica
        with _lib.launch_trace() as tr:
            res = ops.sv_complex(_dev(torch, g[f"{tag}_re"].astype(planes)), _dev(torch, g[f"{tag}_im"].astype(planes)),
                                 _dev(torch, cc), cal_type=cal, **kw)
        if wf == "BB":  # the kernel asked for is the one that ran
            assert any("fft" in k for k in tr.kernels) == (method == "fft"), tr.kernels
        got, exp = res["out"].cpu().numpy(), g[f"{tag}_{cal}"]
        np.testing.assert_array_equal(res["echo_range"].cpu().numpy(), g[f"{tag}_echo_range"])
        np.testing.assert_array_equal(np.isnan(got), np.isnan(exp))
        ok = ~np.isnan(exp)
        if wf == "CW":
            np.testing.assert_allclose(got[ok], exp[ok], rtol=1e-11, atol=1e-10)
        else:
            # the reference's pulse-compressed samples are complex64 (ek80_complex.py:304): judged in linear received
            # power with the peak-relative error model of tests/bb_tolerance.py.  The received power is recovered from
            # the golden dB values by taking the range terms out again (they cancel in got - exp; only the ratio to the
            # ping's peak matters): exp - n log10 R' - 2 alpha R', R' = echo_range - c tau / 4.
            from bb_tolerance import assert_bb_close

            rt = g[f"{tag}_echo_range"] - (cw * tau / 4)[:, :, None]
            with np.errstate(invalid="ignore", divide="ignore"):
                tl = (20.0 if cal == "Sv" else 40.0) * np.log10(rt) + 2 * g[f"{tag}_absorption"][:, :, None] * rt
                prx = np.where(ok & (rt > 0), 10.0 ** ((exp - tl) / 10.0), np.nan)
            assert_bb_close(got, exp, "float64", prx=prx)
            peak = np.nanmax(np.where(ok, exp, -np.inf), axis=2, keepdims=True)
            strong = ok & (exp > peak - 60)
            assert np.abs(got[strong] - exp[strong]).max() < 2e-4


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["float64", "float32"])
@pytest.mark.parametrize("S", [1000, 2052, 260])
@pytest.mark.parametrize("ss_jitter", [0.0, 8.0])
def test_mvbs_of_sv_through_coefficient_rows_fast_kernel(env, dtype, S, ss_jitter):
    """epa_mvbs with coefficient rows in place of the range array (compute_MVBS after a lazy echo_range): the
    specialised kernels (mvbs_of_sv_fixed_kernel + mvbs_of_sv_rows_kernel, asserted through the launch trace) == the
    generic reduction (EPA_NO_FAST_PATH) == the reduction on the echo_range array K1 writes (EPA_BIN_RANGE_AS_STORED),
    partial sums and counts included; NaN Sv skipped, pings before the first edge and empty time bins, range bins finer
    than a sample step.  ss_jitter: a sound speed that moves by metres per second from ping to ping, so that (nearly)
    every column crosses range-bin edges inside a time bin -- the "loose" columns the fixed-bin kernel redoes with its
    lanes over the pings, in S = 260 / 2052 by the TWO lanes of the row's last wavefront (round-3 ADVICE: a stride of 64
    there dropped the pings of the lanes that do not exist).  Time bins of 7 pings: the planner keeps time bins of up to
    8 pings whole (block_reduce.hip make_plan), which is what the specialised kernels serve."""
    import os
    from echopype_amd import _lib
    torch, ops, synth = env
    P = 157
    d = synth.ek60_numpy(3, P, S, ss_every=5 if not ss_jitter else 1)
    if ss_jitter:
        d["sound_speed_indicative"] = d["sound_speed_indicative"] + ss_jitter * np.random.default_rng(11).random((1, P))
    d["backscatter_r"][1, 40:44] = np.nan
    cf = _coef_ek60(torch, ops, d, "Sv")
    dt = getattr(torch, dtype)
    sv, rng = ops.sv_power(_dev(torch, d["backscatter_r"]), cf, dtype=dt)
    ns = torch.from_numpy(d["ping_time"].astype("datetime64[ns]").astype(np.int64)).cuda()
    e0 = int(ns[7])                         # the first 7 pings lie before the first edge
    n_t = 25
    dtn = (int(ns[-1]) - e0) // 21 + 1      # 7 pings per time bin; the last three time bins are empty
    bs = ops.time_bin_offsets(ns, e0, dtn, n_t)
    for rbin in (1.0, 0.07):
        n_r = int(float(np.nanmax(rng.cpu().numpy())) / rbin) + 1
        with _lib.launch_trace() as tr:
            fast = ops.mvbs(sv, bs, n_t, rbin, n_r, coef=cf, coef_as_stored=True, want_partials=True)
        assert "mvbs_of_sv_fixed_kernel" in tr.kernels and "mvbs_of_sv_rows_kernel" in tr.kernels, tr.kernels
        os.environ["EPA_NO_FAST_PATH"] = "1"
        try:
            with _lib.launch_trace() as tr:
                slow = ops.mvbs(sv, bs, n_t, rbin, n_r, coef=cf, coef_as_stored=True, want_partials=True)
        finally:
            del os.environ["EPA_NO_FAST_PATH"]
        assert "mvbs_of_sv_fixed_kernel" not in tr.kernels, tr.kernels
        arr = ops.mvbs(sv, bs, n_t, rbin, n_r, range=rng, want_partials=True)
        for other in (slow, arr):
            np.testing.assert_array_equal(fast["cnt"].cpu().numpy(), other["cnt"].cpu().numpy())
            a, b = fast["MVBS"].cpu().numpy(), other["MVBS"].cpu().numpy()
            np.testing.assert_array_equal(np.isnan(a), np.isnan(b))
            np.testing.assert_allclose(a, b, rtol=1e-12 if dtype == "float64" else 1e-5, atol=1e-12 if dtype == "float64" else 1e-4)
        assert np.isfinite(fast["MVBS"].cpu().numpy()).any() and (fast["cnt"].cpu().numpy()[:, -3:] == 0).all()


# ---- launch sites the API paths reach only from other processes / through other routes (kernel coverage, round 4) ---------
def test_noise_finalize_edge_prepare_affine_rows_first_not_le(env):
    """epa_noise_finalize (merged (sum, count) rows of a noise block cut by a shard edge -> clean/api.py:402-422: mean ->
    dB -> min over the range blocks -> clamp), epa_edge_prepare_max (NaN -> -inf before an all-reduce(MAX)),
    epa_affine_rows (consolidate/api.py:226 on an echo_range ARRAY) and epa_first_not_le on an empty array: in the product
    they run inside the rank processes of the sharded tests or behind other routes; here each against NumPy."""
    torch, ops, synth = env
    rng = np.random.default_rng(8)
    # noise finalize
    rows, Sb = 7, 41
    ssum = rng.random((rows, Sb)) * 1e-7
    cnt = rng.integers(0, 5, (rows, Sb)).astype(np.float64)
    cnt[3] = 0.0                                          # a row with nothing in it: NaN
    ssum[5, 4] = 0.0                                      # an all-zero block mean: -inf dB is the minimum
    for nmax in (float("nan"), -75.0):
        got = ops.noise_finalize(_dev(torch, ssum), _dev(torch, cnt), nmax).cpu().numpy()
        with np.errstate(divide="ignore", invalid="ignore"):
            db = np.where(cnt > 0, 10 * np.log10(ssum / np.where(cnt > 0, cnt, 1.0)), np.nan)
        exp = np.array([np.nanmin(r) if np.isfinite(r).any() or np.isinf(r).any() else np.nan for r in db])
        if nmax == nmax:  # noise.where(noise < max, max) (clean/api.py:418-422): a NaN block becomes the maximum, too
            with np.errstate(invalid="ignore"):
                exp = np.where(exp < nmax, exp, nmax)
        np.testing.assert_array_equal(np.isnan(got), np.isnan(exp))
        np.testing.assert_allclose(got[~np.isnan(exp)], exp[~np.isnan(exp)], rtol=1e-13)
    # NaN -> -inf in place, numbers untouched
    t = _dev(torch, np.array([np.nan, 3.5, -np.inf, np.nan, 0.0]))
    assert ops.edge_prepare_max(t) is t
    np.testing.assert_array_equal(t.cpu().numpy(), [-np.inf, 3.5, -np.inf, -np.inf, 0.0])
    # depth = offset + scale * echo_range per (channel, ping)
    C, P, S = 2, 5, 33
    x = rng.random((C, P, S)) * 100
    x[1, 2, 7:] = np.nan
    sc, off = rng.random((C, P)) + 0.5, rng.random((C, P)) * 10
    for dt in ("float64", "float32"):
        got = ops.affine_rows(_dev(torch, x.astype(dt)), _dev(torch, sc), _dev(torch, off)).cpu().numpy()
        exp = (off[..., None] + sc[..., None] * x.astype(dt).astype(np.float64)).astype(dt)
        np.testing.assert_allclose(got, exp, rtol=1e-15 if dt == "float64" else 1e-6, equal_nan=True)
    # first index whose value is not <= the limit: an empty array answers 0 (= its length)
    import ctypes
    from echopype_amd import _lib
    live, out = _dev(torch, np.arange(4.0)), torch.full((1,), 7, dtype=torch.int64, device="cuda")
    _lib.call("epa_first_not_le", ctypes.c_void_p(live.data_ptr()), 0, 1.0, _lib.F64, ctypes.c_void_p(out.data_ptr()), None)
    torch.cuda.synchronize()
    assert int(out.item()) == 0                                   # n = 0 on a live buffer
    assert int(ops.first_not_le(_dev(torch, np.array([0.5, 1.0, 1.5, 0.2])), 1.0)) == 2


def test_range_power_rows_kernel_serves_what_the_pieces_do_not(env):
    """epa_range_power: S = 1001 takes no 16-byte accesses -- the strided-rows kernel, the same values as the one-piece
    kernel writes for the first 1000 samples of the same rows (bit for bit), NaN where the raw sample is."""
    torch, ops, synth = env
    from echopype_amd import _lib

    d = synth.ek60_numpy(2, 9, 1001)
    coef = _coef_ek60(torch, ops, d, "Sv")
    raw = _dev(torch, d["backscatter_r"])
    for dtype in (torch.float64, torch.float32):
        with _lib.launch_trace() as tr:
            a = ops.range_power(raw, coef, dtype=dtype)
        assert "range_power_kernel" in tr.kernels and "rows_piece_kernel" not in tr.kernels
        with _lib.launch_trace() as tr:
            b = ops.range_power(raw[:, :, :1000].contiguous(), coef, dtype=dtype)
        assert "rows_piece_kernel" in tr.kernels
        a, b = a.cpu().numpy(), b.cpu().numpy()
        np.testing.assert_array_equal(a[:, :, :1000], b)
        np.testing.assert_array_equal(np.isnan(a), np.isnan(d["backscatter_r"]))
