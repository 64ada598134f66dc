"""First contact with a multi-GPU box is a TEST (round-3 review, item 2).

One worker exercises everything the N > 1 bench line and the sharded entry points do --
  (1) bench.Cfg5's ops-level harness: round-robin tiles, every tile edge cutting a 20-s bin, ONE all-reduce of the cut
      bins' (sum, count) rows per pass;
  (2) bench.Cfg5.api_layout: ``sharding.compute_Sv_MVBS(echodata_shard, shard=MVBSShard())`` per resident tile (dataset
      j = tiles j N .. j N + N - 1), results read one tile late;
  (3) ``sharding.{compute_Sv_MVBS, remove_background_noise, compute_MVBS}`` on a contiguous ping split of one file
-- and the parent holds what the ranks report to the ORACLE on the whole volume (commongrid/utils.py:614-627 sums a bin
over all its pings; clean/api.py:402-411 takes a block mean over all its pings).

It runs twice: ranks sharing cuda:0 over gloo (any GPU box: the logic), and -- whenever ``torch.cuda.device_count() >=
2`` -- one NCCL (= RCCL over xGMI) rank per device, with 2 ranks and with every device of the node; on a one-GPU box the
RCCL cases skip.  The driver's scaling bench is then not the first time RCCL sees two ranks."""
import os
import socket

import numpy as np
import pytest

from oracle import calibrate as ocal
from oracle import clean as oclean
from oracle import commongrid as ogrid

pytestmark = pytest.mark.gpu
C, S, TILE_P = 3, 512, 500
OFFSET_NS = 10_000_000_000
PARAMS = ("sample_interval", "transmit_duration_nominal", "transmit_power", "sound_speed_indicative",
          "absorption_indicative")
TABLES = ("gain_correction", "sa_correction", "pulse_length", "equivalent_beam_angle", "frequency_nominal")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _slice_ek60(d, p0, p1):
    out = {}
    P = d["backscatter_r"].shape[1]
    for k, v in d.items():
        if isinstance(v, np.ndarray) and v.ndim >= 2 and v.shape[1] == P and k not in TABLES:
            out[k] = np.ascontiguousarray(v[:, p0:p1])
        elif k == "ping_time":
            out[k] = v[p0:p1]
        else:
            out[k] = v
    return out


def _worker(rank, world, port, backend, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import argparse
    import logging

    import torch
    import torch.distributed as dist

    dev = rank if backend == "nccl" else 0
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    import echopype_amd as ep
    from echopype_amd import sharding

    logging.disable(logging.WARNING)
    out = {"rank": rank, "device": f"cuda{torch.cuda.current_device()}", "comm": sharding._comm_device().type}
    # ---- (1) the ops-level harness of the bench on two datasets' worth of tiles
    args = argparse.Namespace(dtype="float64", steps=1, warmup=0, passes=None, ss_every=3, backend=backend)
    ctx = bench.Ctx(args, world, rank)
    job = bench.Cfg5(ctx, C, TILE_P * 2 * world, S, tile_pings=TILE_P, ss_every=3)
    assert job.gtiles == [rank, rank + world] and job.n_tiles == 2 * world
    info, plan, mv, one_pass = job.layout(OFFSET_NS)
    assert plan.shared and (plan._buf.is_cuda and (plan._hbuf is None) == (backend == "nccl"))
    one_pass(None)
    one_pass(None)  # (a second pass on the kept plan)
    torch.cuda.synchronize()
    owner = {(k, w): o for k, w, _, o in plan.edges}
    ops_bins = []
    for i, m in enumerate(mv):
        _, _, f, l, n = info[i]
        lo = 1 if owner.get((i, 0)) is False else 0
        hi = n - 1 if owner.get((i, 1)) is False else n
        ops_bins.append((np.arange(f + lo, f + hi), m[:, lo:hi].cpu().numpy()))
    out["ops_bins"], out["n_r"], out["e0"] = ops_bins, job.n_r, info[0][1] - info[0][2] * job.BIN_NS
    out["tiles"] = [(g, {k: d[k].cpu().numpy() for k in ("backscatter_r",) + PARAMS}) for g, d in zip(job.gtiles, job.tiles)]
    out["tables"] = {k: job.tiles[0][k].cpu().numpy() for k in TABLES}
    out["ping_time"] = [d["ping_time"] for d in job.tiles]
    # ---- (2) the same tiles through the sharded product entry point, as bench.py's N > 1 headline drives it
    one_pass_api, finish, state, eds = job.api_layout(OFFSET_NS)
    got = []
    import echopype_amd.sharding as sh
    real = sh.compute_Sv_MVBS

    def spy(*a, **k):
        r = real(*a, **k)
        got.append(r)
        return r

    sh.compute_Sv_MVBS = spy
    try:
        one_pass_api(None)
        one_pass_api(None)
        finish()
    finally:
        sh.compute_Sv_MVBS = real
    assert state["n_read"] == 4 and len(got) == 4
    out["api"] = [(np.asarray(mvd["ping_time"].values), np.asarray(mvd["Sv"].values), np.asarray(mvd["echo_range"].values),
                   np.asarray(dsd["Sv"].values[:, :7])) for dsd, mvd in got[2:]]
    out["ranks"] = bench.ranks_info(ctx)
    del got, eds, job, mv, plan
    # ---- (3) a contiguous ping split of ONE file through the three sharded entry points
    P = 96 * world + 37
    d = ep.synth.ek60_numpy(C, P, 600, seed=77)
    d["ping_time"] = d["ping_time"] + np.timedelta64(7, "s")
    p0, p1 = sharding.shard_bounds(P, world, rank)
    ed = ep.echodata.from_ek60_arrays(_slice_ek60(d, p0, p1)).to_device()
    tau0 = d["transmit_duration_nominal"][:, 0]
    shard = sharding.MVBSShard()
    for _ in range(2):  # (the second call finds its plan)
        ds_Sv, mv1 = sharding.compute_Sv_MVBS(ed, range_bin="2m", ping_time_bin="20s", shard=shard,
                                              tau_effective_first_ping=None if rank == 0 else tau0)
    ds = ep.calibrate.compute_Sv(ed)
    sharding.remove_background_noise(ds, 20, 50, ping_offset=p0, background_noise_max="-100.0dB")
    corrected = ds.copy()
    corrected["Sv"] = ds["Sv_corrected"]
    mv2 = sharding.compute_MVBS(corrected, range_bin="2m", ping_time_bin="20s")
    out["split"] = dict(p0=p0, p1=p1, mv=np.asarray(mv1["Sv"].values), t=np.asarray(mv1["ping_time"].values),
                        r=np.asarray(mv1["echo_range"].values), sc=np.asarray(ds["Sv_corrected"].values),
                        mv2=np.asarray(mv2["Sv"].values), t2=np.asarray(mv2["ping_time"].values), plans=len(shard._plans))
    q.put(out)
    dist.barrier()
    dist.destroy_process_group()


def _oracle_sv(h, tables, tau_eff):
    gain = ocal.vend_cal_params_power(h["transmit_duration_nominal"], tables["pulse_length"], tables["gain_correction"])
    sa = ocal.vend_cal_params_power(h["transmit_duration_nominal"], tables["pulse_length"], tables["sa_correction"])
    return ocal.cal_power_ek(
        h["backscatter_r"], sonar="EK60", cal_type="Sv", sample_interval=h["sample_interval"],
        sound_speed=h["sound_speed_indicative"], absorption=h["absorption_indicative"],
        transmit_power=h["transmit_power"], tau_nominal=h["transmit_duration_nominal"], gain=gain, sa_correction=sa,
        psi=tables["equivalent_beam_angle"], f_nominal=tables["frequency_nominal"], tau_eff=tau_eff)


def _close(got, exp, tol=1e-9):
    assert got.shape == exp.shape, (got.shape, exp.shape)
    np.testing.assert_array_equal(np.isnan(got), np.isnan(exp))
    f = np.isfinite(exp)
    assert f.any() and np.max(np.abs(got[f] - exp[f]) / np.maximum(np.abs(exp[f]), 1.0)) < tol


def _run(world, backend):
    import torch
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, backend, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=900) for _ in range(world)), key=lambda o: o["rank"])
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0
    # the ranks were where they should be, and the process group saw all of them
    assert [o["device"] for o in res] == [f"cuda{r if backend == 'nccl' else 0}" for r in range(world)]
    assert all(o["comm"] == ("cuda" if backend == "nccl" else "cpu") for o in res)
    info = res[0]["ranks"]
    assert info["world_size"] == world and info["backend"] == backend and len(info["devices"]) == world
    assert info["devices"] == [f"{r}:cuda{r if backend == 'nccl' else 0}" for r in range(world)]

    # ---- the whole volume, in time order, for the oracle
    tables = res[0]["tables"]
    tiles = dict(t for o in res for t in o["tiles"])
    times = {}
    for o in res:
        for (g, _), t in zip(o["tiles"], o["ping_time"]):
            times[g] = t + np.timedelta64(OFFSET_NS, "ns")
    order = sorted(tiles)
    assert order == list(range(2 * world))
    whole = {k: np.concatenate([tiles[g][k] for g in order], axis=1) for k in ("backscatter_r",) + PARAMS}
    t_all = np.concatenate([times[g] for g in order])
    sv, er = _oracle_sv(whole, tables, np.full(C, 1.024e-3))
    n_r = res[0]["n_r"]
    # (1) ops level: every 20-s bin of the 2 N tiles exactly once, each equal to the oracle's mean over ALL its pings
    exp = ogrid.groupby_mean(sv, er, t_all, ogrid.ping_edges(t_all, "20s"), np.arange(0, n_r + 1.0, 1.0))
    rows = {}
    for o in res:
        for ids, m in o["ops_bins"]:
            for j, b in enumerate(ids):
                assert int(b) not in rows, f"bin {b} reported twice"
                rows[int(b)] = m[:, j]
    first = min(rows)
    assert sorted(rows) == list(range(first, first + exp.shape[1]))
    _close(np.stack([rows[b] for b in sorted(rows)], axis=1), exp)
    # (2) the sharded entry point per tile: dataset j = tiles j N .. j N + N - 1, on ITS OWN grid (a bin cut by the edge
    # between two datasets stays two partial bins, as with the reference run per file)
    for j in range(2):
        a, b = j * world * TILE_P, (j + 1) * world * TILE_P
        exp_j, t_left, r_left = ogrid.compute_MVBS(sv[:, a:b], er[:, a:b], t_all[a:b], "1m", "20s")
        parts = sorted((o["api"][j] for o in res), key=lambda x: x[0][0] if x[0].size else np.datetime64("NaT"))
        t_got = np.concatenate([p[0] for p in parts])
        np.testing.assert_array_equal(t_got, t_left)  # every bin once, in order, over the ranks
        for p in parts:
            np.testing.assert_array_equal(p[2], r_left)  # the range grid of the whole dataset on every rank
        _close(np.concatenate([p[1] for p in parts], axis=1), exp_j)
    for o in res:  # Sv of a rank's tile of dataset 1 (first pings)
        g = o["rank"] + world
        _close(o["api"][1][3], sv[:, g * TILE_P:g * TILE_P + 7])
    # (3) one file split by ping_time
    import echopype_amd as ep  # (host-side synth only)
    P = 96 * world + 37
    d = ep.synth.ek60_numpy(C, P, 600, seed=77)
    d["ping_time"] = d["ping_time"] + np.timedelta64(7, "s")
    tb = {k: d[k] for k in TABLES}
    sv3, er3 = _oracle_sv(d, tb, d["transmit_duration_nominal"][:, 0])
    exp_mv, t_left, r_left = ogrid.compute_MVBS(sv3, er3, d["ping_time"], "2m", "20s")
    sp = [o["split"] for o in res]
    assert [s["p0"] for s in sp] == sorted(s["p0"] for s in sp) and sp[-1]["p1"] == P and all(s["plans"] == 1 for s in sp)
    np.testing.assert_array_equal(np.concatenate([s["t"] for s in sp]), t_left)
    _close(np.concatenate([s["mv"] for s in sp], axis=1), exp_mv)
    for s in sp:
        np.testing.assert_array_equal(s["r"], r_left)
    _, exp_c = oclean.remove_background_noise(sv3, er3, d["absorption_indicative"], 20, 50, "-100.0dB", "3.0dB")
    _close(np.concatenate([s["sc"] for s in sp], axis=1), exp_c)
    exp_mv2, t2, _ = ogrid.compute_MVBS(exp_c, er3, d["ping_time"], "2m", "20s")
    np.testing.assert_array_equal(np.concatenate([s["t2"] for s in sp]), t2)
    _close(np.concatenate([s["mv2"] for s in sp], axis=1), exp_mv2)


@pytest.mark.parametrize("world", [2, 3])
def test_ranks_sharing_one_device_over_gloo(world):
    import torch

    if not torch.cuda.is_available():
        pytest.fail("these tests need a GPU")
    _run(world, "gloo")


@pytest.mark.parametrize("which", ["two", "all"])
def test_rccl_one_rank_per_device(which):
    """Runs by itself the first time a box shows two or more devices; skipped on one."""
    import torch

    if not torch.cuda.is_available():
        pytest.fail("these tests need a GPU")
    n = torch.cuda.device_count()
    if n < 2:
        pytest.skip(f"{n} GPU visible: RCCL with N > 1 ranks needs N devices")
    if which == "all" and n == 2:
        pytest.skip("two devices: covered by the two-rank case")
    _run(2 if which == "two" else n, "nccl")
