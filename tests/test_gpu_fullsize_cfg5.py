"""BASELINE configs[4] at its FULL size on one GPU -- EK60 4 ch x 2 M pings x 4096 range as eight resident tiles of
250 000 pings, exactly what ``bench.py`` times: the ops-level harness (bench.Cfg5.layout: every tile edge cutting a 20-s
time bin, the edge-bin exchange on the path) AND the headline route through the product entry points
(bench.Cfg5.echodata + calibrate.compute_Sv -> commongrid.compute_MVBS per tile, results read one tile late) -- windows
of the first tile, a middle one, the last tile and the tile edges against the oracle.  The oracle cannot run 32.8 G
samples; a 1000-ping window (50 whole time bins, or 51 cut ones) takes it a second.  Needs ~170 GB of HBM."""
import argparse

import numpy as np
import pytest

from oracle import calibrate as ocal
from oracle import commongrid as ogrid

pytestmark = pytest.mark.gpu
C, P_TOTAL, S = 4, 2_000_000, 4096
OFFSET_NS = 10_000_000_000


@pytest.fixture(scope="module")
def job():
    import gc

    import torch

    if not torch.cuda.is_available():
        pytest.fail("needs a GPU")
    gc.collect()
    torch.cuda.empty_cache()
    if torch.cuda.mem_get_info()[0] < 180 * 2**30:
        pytest.skip("needs ~180 GB of free HBM")
    import bench

    ctx = bench.Ctx(argparse.Namespace(dtype="float64", steps=1, warmup=0, passes=None, ss_every=1), 1, 0)
    j = bench.Cfg5(ctx, C, P_TOTAL, S, ss_every=1)
    info, plan, mv, one_pass = j.layout(OFFSET_NS)
    assert len(j.tiles) == 8 and plan.shared and len(plan.edges) == 14 and not j.keep_all
    one_pass(None)
    torch.cuda.synchronize()
    yield dict(torch=torch, ctx=ctx, job=j, info=info, mv=mv)
    del j, mv, info, plan, one_pass
    gc.collect()
    torch.cuda.empty_cache()


def _oracle_window(job, tile, a, w):
    d = job["job"].tiles[tile]
    h = {k: d[k][:, a:a + w].cpu().numpy() for k in ("backscatter_r", "sample_interval", "transmit_duration_nominal",
                                                     "transmit_power", "sound_speed_indicative", "absorption_indicative")}
    for k in ("gain_correction", "sa_correction", "pulse_length", "equivalent_beam_angle", "frequency_nominal"):
        h[k] = d[k].cpu().numpy()
    gain = ocal.vend_cal_params_power(h["transmit_duration_nominal"], h["pulse_length"], h["gain_correction"])
    sa = ocal.vend_cal_params_power(h["transmit_duration_nominal"], h["pulse_length"], h["sa_correction"])
    sv, er = ocal.cal_power_ek(
        h["backscatter_r"], sonar="EK60", cal_type="Sv", sample_interval=h["sample_interval"],
        sound_speed=h["sound_speed_indicative"], absorption=h["absorption_indicative"],
        transmit_power=h["transmit_power"], tau_nominal=h["transmit_duration_nominal"], gain=gain, sa_correction=sa,
        psi=h["equivalent_beam_angle"], f_nominal=h["frequency_nominal"], tau_eff=np.full(C, 1.024e-3))
    t = d["ping_time"][a:a + w] + np.timedelta64(OFFSET_NS, "ns")
    return sv, er, t


def _close(got, exp, tol=1e-9):
    np.testing.assert_array_equal(np.isnan(got), np.isnan(exp))
    f = np.isfinite(exp)
    assert np.max(np.abs(got[f] - exp[f]) / np.maximum(np.abs(exp[f]), 1.0)) < tol


@pytest.mark.parametrize("tile,a", [(0, 10), (0, 123_450), (7, 249_000 - 10), (3, 77_770)])
def test_cfg5_tile_windows_of_the_mvbs_match_the_oracle(job, tile, a):
    """1000 pings starting 10 pings into a bin... the window [a, a + 1000) with a = 10 (mod 20) covers exactly the 50
    whole bins behind the tile's (cut) first bin."""
    assert a % 20 == 10
    sv, er, t = _oracle_window(job, tile, a, 1000)
    n_r = job["job"].n_r
    exp = ogrid.groupby_mean(sv, er, t, ogrid.ping_edges(t, "20s"), np.arange(0, n_r + 1.0, 1.0))
    assert exp.shape[1] == 50
    b0 = (a + 10) // 20   # local bin of the window's first ping (tile starts 10 s into global bin `first`)
    _close(job["mv"][tile][:, b0:b0 + 50].cpu().numpy(), exp)


def test_cfg5_last_tile_sv_window_matches_the_oracle(job):
    """Sv of every tile goes through ONE reused buffer at N = 1: after a pass it holds the last tile."""
    sv, er, _ = _oracle_window(job, 7, 200_000, 600)
    _close(job["job"].sv[7][:, 200_000:200_600].cpu().numpy(), sv)


@pytest.mark.parametrize("edge", [0, 3, 6])
def test_cfg5_time_bins_cut_by_a_tile_edge_hold_all_their_pings(job, edge):
    """The bin cut by the edge between tiles ``edge`` and ``edge + 1``: 10 pings on either side.  The owner (the
    earlier tile) reports it from the exchanged totals; it must equal the oracle's mean over all 20 pings."""
    sv_a, er_a, t_a = _oracle_window(job, edge, 250_000 - 10, 10)
    sv_b, er_b, t_b = _oracle_window(job, edge + 1, 0, 10)
    sv, er, t = np.concatenate((sv_a, sv_b), axis=1), np.concatenate((er_a, er_b), axis=1), np.concatenate((t_a, t_b))
    n_r = job["job"].n_r
    exp = ogrid.groupby_mean(sv, er, t, ogrid.ping_edges(t, "20s"), np.arange(0, n_r + 1.0, 1.0))
    assert exp.shape[1] == 1
    _close(job["mv"][edge][:, -1:].cpu().numpy(), exp)
    # ... and it differs from what either side alone would report (the exchange did something)
    half = ogrid.groupby_mean(sv_a, er_a, t_a, ogrid.ping_edges(t, "20s"), np.arange(0, n_r + 1.0, 1.0))
    assert np.nanmax(np.abs(half - exp)) > 1e-6


def test_cfg5_tiles_through_the_reference_calls_match_the_oracle(job):
    """The headline route of bench.py at N = 1: every resident tile as an EchoData through calibrate.compute_Sv and
    commongrid.compute_MVBS(ds, "1m", "20s"), the next tile launched before the previous result is read (the calls do
    not wait for the GPU; reading a deferred MVBS dataset does).  Each tile is its own dataset, as with the reference
    run per file: its first and last bin hold the 10 pings on its side of the tile edge.  Windows of three tiles: 50
    whole bins of the MVBS, the cut first bin, 600 pings of the Sv array."""
    import collections
    import logging

    import echopype_amd as ep
    from echopype_amd.xr_lite import DeferredDataset

    torch, j = job["torch"], job["job"]
    j.sv = None            # (the ops-level harness' reused Sv buffer: the API allocates its own outputs)
    job["mv"].clear()
    torch.cuda.empty_cache()
    logging.disable(logging.WARNING)
    checked = []

    def check(tile, ds, mv):
        assert mv["Sv"].shape == (C, 12501, j.n_r) and tuple(mv["Sv"].dims) == ("channel", "ping_time", "echo_range")
        t0 = np.asarray(mv["ping_time"].values)[0]
        d = j.tiles[tile]
        first_ping = d["ping_time"][0] + np.timedelta64(OFFSET_NS, "ns")
        assert first_ping - t0 == np.timedelta64(10, "s")          # the tile starts 10 s into its first bin
        got = mv["Sv"].data.tensor
        for a in (10, 123_450, 249_000 - 10):
            sv, er, t = _oracle_window(job, tile, a, 1000)
            exp = ogrid.groupby_mean(sv, er, t, ogrid.ping_edges(t, "20s"), np.arange(0, j.n_r + 1.0, 1.0))
            b0 = (a + 10) // 20
            _close(got[:, b0:b0 + 50].cpu().numpy(), exp)
        sv, er, t = _oracle_window(job, tile, 0, 10)               # the cut first bin: this tile's 10 pings only
        exp = ogrid.groupby_mean(sv, er, t, ogrid.ping_edges(t, "20s"), np.arange(0, j.n_r + 1.0, 1.0))
        _close(got[:, :1].cpu().numpy(), exp)
        sv, er, _ = _oracle_window(job, tile, 200_000, 600)
        _close(ds["Sv"].data.tensor[:, 200_000:200_600].cpu().numpy(), sv)
        np.testing.assert_array_equal(mv["echo_range"].values, np.arange(j.n_r, dtype=np.float64))
        checked.append(tile)

    try:
        pending = collections.deque()
        for tile in range(len(j.tiles)):
            ed = j.echodata(tile, OFFSET_NS)
            ds = ep.calibrate.compute_Sv(ed)
            mv = ep.commongrid.compute_MVBS(ds, range_bin="1m", ping_time_bin="20s")
            assert isinstance(mv, DeferredDataset) and not mv.resolved
            pending.append((tile, ds, mv))
            while len(pending) > 1:
                t_, ds_, mv_ = pending.popleft()
                if t_ in (0, 3):
                    check(t_, ds_, mv_)
                else:
                    mv_["Sv"].shape
                del ds_, mv_
        t_, ds_, mv_ = pending.popleft()
        check(t_, ds_, mv_)
    finally:
        logging.disable(logging.NOTSET)
    assert checked == [0, 3, 7]
