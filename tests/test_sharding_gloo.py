"""N>1 path on CPU: world_size-2 / -3 gloo processes exercise the cross-shard bookkeeping of
echopype_amd.sharding (global time grid, range-grid max, the EdgeExchange behind the straddling MVBS time bins and
the straddling background-noise ping blocks, several resident segments per rank) against a single-process NumPy
evaluation.  The per-sample partial sums come from NumPy here (the HIP kernels need a GPU: tests/test_gpu_sharded.py
runs the same exchange on their output)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import commongrid as ogrid


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _partials(Sv, er, ns, e0, dt, n_t, first, r_edges):
    """NumPy (sum, cnt) per (channel, local time bin, range bin) -- what the kernels produce."""
    C = Sv.shape[0]
    nr = len(r_edges) - 1
    ssum = np.zeros((C, n_t, nr))
    cnt = np.zeros((C, n_t, nr), dtype=np.int64)
    tb = (ns - e0) // dt - first
    lin = 10 ** (Sv / 10)
    for c in range(C):
        rb = ogrid.bin_index(er[c], r_edges)
        ok = (rb >= 0) & ~np.isnan(lin[c])
        flat = (tb[:, None] * nr + rb)[ok]
        ssum[c] = np.bincount(flat, weights=lin[c][ok], minlength=n_t * nr).reshape(n_t, nr)
        cnt[c] = np.bincount(flat, minlength=n_t * nr).reshape(n_t, nr)
    return ssum, cnt


def _worker(rank, world, port, P_total, split, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from echopype_amd import sharding

    rng = np.random.default_rng(42)
    C, S = 2, 64
    Sv = rng.normal(-70, 6, size=(C, P_total, S))
    er = np.tile(np.arange(S) * 0.19, (C, P_total, 1))
    ns = (np.datetime64("2026-05-01T00:00:03", "ns").astype(np.int64) + np.arange(P_total) * 10**9)
    bounds = [0] + list(split) + [P_total]
    p0, p1 = bounds[rank], bounds[rank + 1]
    dt = 20 * 10**9
    e0, n_glob = sharding.global_time_grid(ns[p0:p1], dt)
    rmax = sharding.global_max(float(np.nanmax(er[:, p0:p1])) if p1 > p0 else -np.inf)
    r_edges = np.arange(0, rmax + 1.0, 1.0)
    first, last = sharding.local_bin_span(ns[p0:p1], e0, dt)
    n_t = last - first + 1
    ssum, cnt = _partials(Sv[:, p0:p1], er[:, p0:p1], ns[p0:p1], e0, dt, n_t, first, r_edges)
    ts, tc = torch.from_numpy(ssum), torch.from_numpy(cnt)
    keep = sharding.merge_straddling_bins(ts, tc, first, last)
    with np.errstate(divide="ignore", invalid="ignore"):
        mv = np.where(tc.numpy() > 0, 10 * np.log10(ts.numpy() / np.maximum(tc.numpy(), 1)), np.nan)
    q.put((rank, e0, n_glob, first, keep, mv[:, keep]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("P_total,split", [(100, (47,)), (100, (60,)), (45, (10,)), (30, (8, 14))])
def test_straddling_bins_merge_equals_single_process(P_total, split):
    world = len(split) + 1
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, P_total, split, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process expectation
    rng = np.random.default_rng(42)
    C, S = 2, 64
    Sv = rng.normal(-70, 6, size=(C, P_total, S))
    er = np.tile(np.arange(S) * 0.19, (C, P_total, 1))
    pt = np.datetime64("2026-05-01T00:00:03", "ns") + (np.arange(P_total) * 10**9).astype("timedelta64[ns]")
    exp, t_left, _ = ogrid.compute_MVBS(Sv, er, pt, "1m", "20s")
    assert all(r[1] == t_left[0].astype(np.int64) and r[2] == len(t_left) for r in res)
    # concatenating the kept bins of every rank reproduces the global grid exactly once per bin
    got = np.full_like(exp, np.nan)
    seen = np.zeros(len(t_left), dtype=int)
    for rank, e0, n_glob, first, keep, mv in res:
        ids = first + np.flatnonzero(keep)
        seen[ids] += 1
        got[:, ids] = mv
    assert (seen == 1).all(), seen
    np.testing.assert_allclose(got, exp, rtol=1e-12, equal_nan=True)


def test_shard_bounds_align_to_bins():
    from echopype_amd.sharding import shard_bounds

    spans = [shard_bounds(2_000_000, 8, r, align=20) for r in range(8)]
    assert spans[0] == (0, 250_000) and spans[-1] == (1_750_000, 2_000_000)
    assert all(a % 20 == 0 for a, _ in spans)
    spans = [shard_bounds(1003, 4, r, align=20) for r in range(4)]
    assert spans[-1][1] == 1003 and sum(b - a for a, b in spans) == 1003


# ---- several resident segments (tiles) per rank: intra-rank and inter-rank shared bins in ONE exchange ----------
def _worker_segments(rank, world, port, P_total, cuts, owner_of, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from echopype_amd import sharding

    rng = np.random.default_rng(5)
    C, S = 2, 40
    Sv = rng.normal(-70, 6, size=(C, P_total, S))
    er = np.tile(np.arange(S) * 0.23, (C, P_total, 1))
    ns = (np.datetime64("2026-05-01T00:00:07", "ns").astype(np.int64) + np.arange(P_total) * 10**9)
    bounds = [0] + list(cuts) + [P_total]
    segs = [(bounds[i], bounds[i + 1]) for i in range(len(bounds) - 1) if owner_of[i] == rank]
    dt = 20 * 10**9
    mine = np.concatenate([ns[a:b] for a, b in segs]) if segs else ns[:0]
    e0, _ = sharding.global_time_grid(mine, dt)
    rmax = sharding.global_max(float(np.nanmax(er)))
    r_edges = np.arange(0, rmax + 1.0, 1.0)
    spans, parts = [], []
    for a, b in segs:
        first, last = sharding.local_bin_span(ns[a:b], e0, dt)
        ssum, cnt = _partials(Sv[:, a:b], er[:, a:b], ns[a:b], e0, dt, last - first + 1, first, r_edges)
        spans.append((first, last))
        parts.append((torch.from_numpy(ssum), torch.from_numpy(cnt.astype(np.float64))))
    plan = sharding.EdgeExchange(spans, C, len(r_edges) - 1, "cpu")
    rows = {}
    for k, (ssum, cnt) in enumerate(parts):
        for w, r in sharding.mvbs_edge_rows(ssum, cnt).items():
            rows[(k, w)] = r
    tot = plan.merge(rows)
    out = []
    for k, (ssum, cnt) in enumerate(parts):
        keep = np.ones(ssum.shape[1], dtype=bool)
        for kk, w, _, owner in plan.edges:
            if kk != k:
                continue
            j = 0 if w == 0 else ssum.shape[1] - 1
            ssum[:, j], cnt[:, j] = tot[(k, w)]
            keep[j] = owner
        with np.errstate(divide="ignore", invalid="ignore"):
            mv = np.where(cnt.numpy() > 0, 10 * np.log10(ssum.numpy() / np.maximum(cnt.numpy(), 1)), np.nan)
        out.append((spans[k][0], keep, mv[:, keep]))
    q.put((rank, e0, out))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("P_total,cuts,owner_of", [
    (200, (50, 110, 150), (0, 0, 1, 1)),      # two tiles per rank, every cut inside a 20-ping bin
    (120, (30, 45, 100), (0, 1, 1, 0)),       # rank 0 holds the first and the last tile; a 15-ping tile inside one bin
    (90, (20, 40, 60), (0, 1, 2, 0)),         # three ranks, cuts on bin edges but the 7-s phase still splits bins
])
def test_edge_exchange_with_several_segments_per_rank(P_total, cuts, owner_of):
    world = max(owner_of) + 1
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_segments, args=(r, world, port, P_total, cuts, owner_of, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    rng = np.random.default_rng(5)
    C, S = 2, 40
    Sv = rng.normal(-70, 6, size=(C, P_total, S))
    er = np.tile(np.arange(S) * 0.23, (C, P_total, 1))
    pt = np.datetime64("2026-05-01T00:00:07", "ns") + (np.arange(P_total) * 10**9).astype("timedelta64[ns]")
    exp, t_left, _ = ogrid.compute_MVBS(Sv, er, pt, "1m", "20s")
    got = np.full_like(exp, np.nan)
    seen = np.zeros(len(t_left), dtype=int)
    for rank, e0, segs in res:
        assert e0 == t_left[0].astype(np.int64)
        for first, keep, mv in segs:
            ids = first + np.flatnonzero(keep)
            seen[ids] += 1
            got[:, ids] = mv
    assert (seen == 1).all(), seen
    np.testing.assert_allclose(got, exp, rtol=1e-12, equal_nan=True)


# ---- background-noise ping blocks cut by a shard edge (clean/api.py:402-411: block mean BEFORE the min) ----------
def _noise_partials(Sv, er, alpha, ping_num, rsn, phase):
    """NumPy stand-in of epa_noise_estimate(ping_phase, want_edges): noise per local block + raw (sum, count) per
    range block of the first / last block."""
    C, P, S = Sv.shape
    with np.errstate(invalid="ignore", divide="ignore"):
        tl = 20 * np.log10(np.where(er >= 1, er, 1)) + 2 * alpha * er
        lin = 10 ** ((Sv - tl) / 10)
    blk = (np.arange(P) + phase) // ping_num
    nb, Sb = blk.max() + 1, -(-S // rsn)
    rb = np.arange(S) // rsn
    ssum, cnt = np.zeros((C, nb, Sb)), np.zeros((C, nb, Sb))
    ok = ~np.isnan(lin)
    for c in range(C):
        flat = (blk[:, None] * Sb + rb[None, :])
        ssum[c] = np.bincount(flat[ok[c]], weights=lin[c][ok[c]], minlength=nb * Sb).reshape(nb, Sb)
        cnt[c] = np.bincount(flat[ok[c]], minlength=nb * Sb).reshape(nb, Sb)
    with np.errstate(invalid="ignore", divide="ignore"):
        noise = np.nanmin(np.where(cnt > 0, 10 * np.log10(ssum / np.maximum(cnt, 1)), np.nan), axis=2)
    es, ec = np.zeros((2, C, Sb)), np.zeros((2, C, Sb))
    es[0], ec[0] = ssum[:, 0], cnt[:, 0]
    if nb > 1:
        es[1], ec[1] = ssum[:, -1], cnt[:, -1]
    return noise, es, ec


def _worker_noise(rank, world, port, P_total, split, ping_num, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from echopype_amd import sharding

    rng = np.random.default_rng(9)
    C, S, rsn = 2, 50, 8
    Sv = rng.normal(-80, 5, size=(C, P_total, S))
    Sv[rng.random(Sv.shape) < 0.05] = np.nan
    er = np.tile(np.linspace(0, 40, S), (C, P_total, 1))
    bounds = [0] + list(split) + [P_total]
    p0, p1 = bounds[rank], bounds[rank + 1]
    noise, es, ec = _noise_partials(Sv[:, p0:p1], er[:, p0:p1], 0.01, ping_num, rsn, p0 % ping_num)
    noise_t = torch.from_numpy(noise.copy())

    def finalize(s, c):  # NumPy stand-in of epa_noise_finalize
        s, c = s.numpy(), c.numpy()
        with np.errstate(invalid="ignore", divide="ignore"):
            return torch.from_numpy(np.nanmin(np.where(c > 0, 10 * np.log10(s / np.maximum(c, 1)), np.nan), axis=1))

    sharding.merge_noise_edges(noise_t, torch.from_numpy(es), torch.from_numpy(ec), p0, p1 - p0, ping_num, finalize=finalize)
    q.put((rank, p0, p1, noise_t.numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("P_total,split,ping_num", [(100, (47,), 10), (100, (50,), 10), (64, (5, 9), 20), (30, (7, 19), 4)])
def test_noise_blocks_cut_by_shard_edges_equal_single_process(P_total, split, ping_num):
    from oracle import clean as oclean

    world = len(split) + 1
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_noise, args=(r, world, port, P_total, split, ping_num, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    rng = np.random.default_rng(9)
    C, S, rsn = 2, 50, 8
    Sv = rng.normal(-80, 5, size=(C, P_total, S))
    Sv[rng.random(Sv.shape) < 0.05] = np.nan
    er = np.tile(np.linspace(0, 40, S), (C, P_total, 1))
    # the single-process oracle: Sv_noise - TL is the per-ping noise of the whole dataset
    sn = oclean.estimate_background_noise(Sv, er, 0.01, ping_num, rsn)
    with np.errstate(invalid="ignore", divide="ignore"):
        tl = 20 * np.log10(np.where(er >= 1, er, 1)) + 2 * 0.01 * er
    per_ping = (sn - tl)[:, :, 3]
    for rank, p0, p1, noise in res:
        blk = (np.arange(p1 - p0) + p0 % ping_num) // ping_num
        np.testing.assert_allclose(noise[:, blk], per_ping[:, p0:p1], rtol=1e-12, atol=1e-10)


# ---- MVBSShard.finish (merge + finalise into the MVBS rows), kept plans, collective decisions ------------------------
def _worker_finish(rank, world, port, P_total, split, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from echopype_amd import sharding

    rng = np.random.default_rng(42)
    C, S = 2, 64
    Sv = rng.normal(-70, 6, size=(C, P_total, S))
    er = np.tile(np.arange(S) * 0.19, (C, P_total, 1))
    ns = (np.datetime64("2026-05-01T00:00:03", "ns").astype(np.int64) + np.arange(P_total) * 10**9)
    bounds = [0] + list(split) + [P_total]
    p0, p1 = bounds[rank], bounds[rank + 1]
    dt = 20 * 10**9
    shard = sharding.MVBSShard()
    outs = []
    for rep in range(2):  # the second call finds the plan kept by the first (one scalar all-reduce checks that on every rank)
        e0, n_glob, first, last = shard.time_grid(ns[p0:p1], dt, "left")
        rmax = shard.range_max(float(np.nanmax(er[:, p0:p1])))
        r_edges = np.arange(0, rmax + 1.0, 1.0)
        n_t = last - first + 1
        ssum, cnt = _partials(Sv[:, p0:p1], er[:, p0:p1], ns[p0:p1], e0, dt, n_t, first, r_edges)
        with np.errstate(divide="ignore", invalid="ignore"):
            mv = np.where(cnt > 0, 10 * np.log10(ssum / np.maximum(cnt, 1)), np.nan)
        res = {"sum": torch.from_numpy(ssum), "cnt": torch.from_numpy(cnt.astype(np.float64)), "MVBS": torch.from_numpy(mv)}
        kept, lo = shard.finish(res, first, last, float("nan"))
        outs.append((first + lo, kept.numpy().copy()))
    assert len(shard._plans) == 1
    np.testing.assert_array_equal(outs[0][1], outs[1][1])
    # a decision taken on one rank only becomes everybody's
    agreed = shard.agree(rank == world - 1)
    q.put((rank, e0, outs[0][0], outs[0][1], agreed))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("P_total,split", [(100, (47,)), (70, (13, 36))])
def test_mvbs_shard_finish_kept_plan_and_collective_decision(P_total, split):
    world = len(split) + 1
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_finish, args=(r, world, port, P_total, split, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    rng = np.random.default_rng(42)
    C, S = 2, 64
    Sv = rng.normal(-70, 6, size=(C, P_total, S))
    er = np.tile(np.arange(S) * 0.19, (C, P_total, 1))
    pt = np.datetime64("2026-05-01T00:00:03", "ns") + (np.arange(P_total) * 10**9).astype("timedelta64[ns]")
    exp, t_left, _ = ogrid.compute_MVBS(Sv, er, pt, "1m", "20s")
    got = np.concatenate([r[3] for r in res], axis=1)  # the ranks' kept bins, in rank order = in time order
    assert [r[2] for r in res] == list(np.cumsum([0] + [r[3].shape[1] for r in res[:-1]]))
    np.testing.assert_allclose(got, exp, rtol=1e-12, equal_nan=True)
    assert all(r[4] for r in res)


# ---- the plan cache is keyed on the GLOBAL layout (round-3 ADVICE) -----------------------------------------------------
def _worker_plan_cache(rank, world, port, q):
    """Rank 1 goes layout 1 -> 2 -> 1 while rank 0 stays on its own: a cache keyed on the local spans alone lets both
    ranks 'hit' on the third call with plans built against different partners (mismatched ``shared`` -> one rank skips
    the all-reduce and the job hangs, or wrong slot groups)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from echopype_amd import sharding

    shard = sharding.ShardContext()
    C, R = 2, 5
    # rank 0 always holds bins 0..4; rank 1 holds 4..9 (bin 4 shared), then 5..9 (nothing shared), then 4..9 again
    seq = [[(0, 4)], [(0, 4)], [(0, 4)], [(0, 4)]] if rank == 0 else [[(4, 9)], [(5, 9)], [(4, 9)], [(5, 9)]]
    out = []
    for step, spans in enumerate(seq):
        plan = shard.plan(spans, C, R, "cpu")
        rows = {}
        for which in (0, 1):
            rows[(0, which)] = (torch.full((C, R), float(10 * rank + which + 1), dtype=torch.float64),
                                torch.full((C, R), 1.0, dtype=torch.float64))
        tot = plan.merge(rows)
        out.append((plan.shared, len(plan.edges), {k: float(v[0][0, 0]) for k, v in tot.items()}))
    # the plan's message doubles as the vote on a fallback: a rank that declines takes every rank with it, in ONE message
    n_plans = len(shard._plans)
    votes = [shard.plan(seq[0], C, R, "cpu", declined=(rank == 1 and k == 1)) for k in range(3)]
    assert [v is None for v in votes] == [False, True, False] and len(shard._plans) == n_plans
    assert votes[0] is shard.plan(seq[0], C, R, "cpu")
    q.put((rank, out, len(shard._plans)))
    dist.barrier()
    dist.destroy_process_group()


def test_plan_cache_is_keyed_on_every_ranks_layout():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_plan_cache, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    r0, r1 = res[0][1], res[1][1]
    # steps 0 and 2: bin 4 is rank 0's LAST edge (value 2) and rank 1's FIRST edge (value 11): total 13 on both
    for step in (0, 2):
        assert r0[step] == (True, 1, {(0, 1): 13.0}) and r1[step] == (True, 1, {(0, 0): 13.0})
    for step in (1, 3):  # nothing shared anywhere: no exchange on either rank
        assert r0[step] == (False, 0, {}) and r1[step] == (False, 0, {})
    assert res[0][2] == 2 and res[1][2] == 2  # two global layouts, two plans on every rank, both re-used


# ---- the time grid and the range cap in one control message ------------------------------------------------------------
def _worker_grid(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from echopype_amd import sharding

    ns = np.datetime64("2026-05-01T00:00:03", "ns").astype(np.int64) + (np.arange(40) + 40 * rank) * 10**9
    shard = sharding.MVBSShard()
    reach = [12.5, float("nan"), 786.4921875][rank]           # rank 1 has no valid range at all
    a = shard.grid(ns, 20 * 10**9, "left", reach)
    b = shard.time_grid(ns, 20 * 10**9, "left") + (shard.range_max(reach),)
    none = shard.grid(ns, 20 * 10**9, "left", float("nan"))  # nobody has one: NaN
    # the same message as the vote on the call's route: the route every rank names, 0 when they differ (a rank whose
    # pings are unsorted names 0 and hands its times over as they are)
    same = shard.grid(ns, 20 * 10**9, "left", reach, sorted_valid=True, route=2)
    mixed = shard.grid(ns[::-1] if rank == 1 else ns, 20 * 10**9, "left", reach if rank != 1 else float("nan"),
                       sorted_valid=rank != 1, route=0 if rank == 1 else 1)
    q.put((rank, a, b, none[4], same, mixed))
    dist.barrier()
    dist.destroy_process_group()


def test_grid_in_one_message_equals_the_two_separate_agreements():
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_grid, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, a, b, none, same, mixed in res:
        assert a == b and a[4] == 786.4921875 and np.isnan(none)
        assert same == a + (2,) and mixed == a + (0,)
    assert [r[1][2] for r in res] == [0, 2, 4] and res[0][1][0] == res[2][1][0]


# ---- whole-file scalars of a ping-sharded file (sharding.file_scalars) ------------------------------------------------
def _worker_file_scalars(rank, world, port, cuts, filter_pings, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from shard_cases import ek80_bb_file, shard_of

    from echopype_amd import sharding

    ed = ek80_bb_file(P=40, S=16, late=12, filter_pings=filter_pings)
    bounds = [0] + list(cuts) + [40]
    fs = sharding.file_scalars(shard_of(ed, bounds[rank], bounds[rank + 1]), waveform_mode="BB", encode_mode="complex")
    q.put((rank, fs))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("cuts,filter_pings", [((8, 20), None), ((8, 20), [0, 12, 25]), ((13,), [0, 12, 25]), ((30,), [0, 5])])
def test_file_scalars_equal_the_whole_files(cuts, filter_pings):
    """Every rank gets the facts of the WHOLE file -- first ping's pulse length, each channel's first valid ping, the
    transmit parameters' (min, max), the filter intervals' starts / first pulse lengths / parameter ranges -- whichever
    shard holds the pings they come from (calibrate/api.py:98-160, calibrate_ek.py:113-162, ek80_complex.py:255-282)."""
    import sys

    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from shard_cases import ek80_bb_file

    from echopype_amd.echodata import BEAM1

    world = len(cuts) + 1
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_file_scalars, args=(r, world, port, cuts, filter_pings, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [fs for _, fs in sorted((q.get(timeout=180) for _ in range(world)), key=lambda t: t[0])]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    beam = ek80_bb_file(P=40, S=16, late=12, filter_pings=filter_pings)[BEAM1]
    tau = np.asarray(beam["transmit_duration_nominal"].values)
    ns = np.asarray(beam["ping_time"].values).astype("datetime64[ns]").view(np.int64)
    for fs in res:
        np.testing.assert_array_equal(fs["tau_nominal_first_ping"], tau[:, 0])  # (NaN for the channel that starts late)
        np.testing.assert_array_equal(fs["first_valid_ping_time"], [ns[0], ns[12]])
        for name, (lo, hi) in fs["transmit_params"].items():
            a = np.asarray(beam[name].values)
            np.testing.assert_array_equal(lo, np.nanmin(a, axis=1))
            np.testing.assert_array_equal(hi, np.nanmax(a, axis=1))
        if filter_pings is None:
            assert "interval_starts" not in fs
            continue
        F = len(filter_pings)
        exp_start = np.array([[not np.isnan(tau[c, p]) for p in filter_pings] for c in range(2)])
        np.testing.assert_array_equal(fs["interval_starts"], exp_start)
        exp_tau0 = np.array([[tau[c, p] for p in filter_pings] for c in range(2)])
        np.testing.assert_array_equal(fs["interval_tau0"], exp_tau0)
        lo, hi = fs["interval_transmit_params"]["transmit_duration_nominal"]
        assert lo.shape == (2, F)
        for c in range(2):
            for f in range(F):
                if exp_start[c, f]:
                    assert lo[c, f] == hi[c, f] == np.nanmin(tau[c])
                else:
                    assert lo[c, f] == np.inf and hi[c, f] == -np.inf


# ---- the control group of a strict SUB-group is created by its members alone (round-4 advice) ---------------------------
def _worker_subgroup(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from echopype_amd import sharding

    sub = dist.new_group(ranks=[0, 1])  # (collective over the world: every rank, also rank 2)
    out = None
    if rank < 2:
        sharding.SEPARATE_CONTROL_GROUP = True  # the path an RCCL data group takes: a gloo group of the same ranks
        ns = np.datetime64("2026-05-01T00:00:03", "ns").astype(np.int64) + (np.arange(30) + 30 * rank) * 10**9
        e0, n = sharding.global_time_grid(ns, 20 * 10**9, group=sub)   # first sharded call: creates the control group
        ctl = sharding.control_group(sub)
        assert ctl is not sub and dist.get_process_group_ranks(ctl) == [0, 1]
        ctx = sharding.ShardContext(sub)
        out = (e0, n, sharding.global_max(float(rank), group=sub), ctx.agree(rank == 1))
        # a control group handed in instead
        mine = dist.new_group(ranks=[0, 1], backend="gloo", use_local_synchronization=True)
        sharding.register_control_group(sub, mine)
        assert sharding.control_group(sub) is mine
        out += (sharding.global_max(10.0 + rank, group=sub),)
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_control_group_of_a_sub_group_is_created_by_its_members_only():
    """Ranks 0 and 1 of a three-rank world shard a dataset between them; rank 2 never calls into sharding.  The gloo
    control group behind the sub-group's host-side agreements must come up without rank 2 (dist.new_group with
    use_local_synchronization) -- it used to wait for every rank of the default group."""
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_subgroup, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[2] is None
    assert res[0] == res[1] and res[0][1] == 4 and res[0][2] == 1.0 and res[0][3] is True and res[0][4] == 11.0
