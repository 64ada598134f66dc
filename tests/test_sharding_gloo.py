"""N>1 path on CPU: world_size-2 gloo processes exercise the cross-shard bookkeeping of
echopype_amd.sharding (global time grid, range-grid max, straddling-bin merge) against a
single-process NumPy evaluation of the same MVBS partial sums."""
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
