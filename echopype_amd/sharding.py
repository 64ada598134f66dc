"""Ping-sharded multi-GPU execution: one process per GPU, ``torch.distributed`` (backend "nccl" =
RCCL over xGMI on ROCm; "gloo" for the CPU tests).

The path shards along ``ping_time``: calibration is independent per ping; the only
cross-shard quantities are
  * scalars fixed before sharding -- EK60 tau_effective (ping 0 of the whole file), the day origin
    of the ping-time bins and the range-bin grid (all-reduce MIN / MAX of one number each);
  * MVBS time bins that straddle a shard edge: each rank contributes the raw linear (sum, count) of
    its FIRST and LAST local time bin, one all-reduce(SUM) over a (2 * world, C, n_rbins) buffer
    (<= a few hundred KB: latency-bound on xGMI, never bandwidth-bound), then the lowest rank that
    holds a shared bin finalises it (10*log10(sum/count)) and the others drop their copy.
When every shard holds a whole number of time bins (the bench's weak-scaling layout) the straddle
exchange is skipped and the data path has no collective at all.

This module holds only the bin bookkeeping and the exchange; kernels are called through ops.
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_bounds(P_total, world, rank, align=1):
    """Contiguous ping block of ``rank``; block edges are multiples of ``align`` where possible."""
    per = -(-P_total // world)
    if align > 1:
        per = -(-per // align) * align
    p0 = min(P_total, rank * per)
    p1 = min(P_total, p0 + per)
    return p0, p1


def global_time_grid(local_ping_ns, dt_ns, group=None):
    """(first_edge, n_bins_global) of the resample grid of the WHOLE dataset from shard-local ping
    times: all-reduce MIN of the first and MAX of the last valid timestamp (two int64)."""
    t = np.asarray(local_ping_ns, dtype=np.int64)
    t = t[t != np.iinfo(np.int64).min]
    lo = torch.tensor([t.min() if t.size else np.iinfo(np.int64).max], dtype=torch.int64)
    hi = torch.tensor([t.max() if t.size else np.iinfo(np.int64).min + 1], dtype=torch.int64)
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dev = _comm_device()
        lo, hi = lo.to(dev), hi.to(dev)
        dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
    first, last = int(lo.item()), int(hi.item())
    day = 86400 * 10**9
    origin = (first // day) * day
    e0 = origin + ((first - origin) // dt_ns) * dt_ns
    return e0, int((last - e0) // dt_ns + 1)


def _comm_device():
    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")


def global_max(value, group=None):
    """all-reduce MAX of one float (e.g. nanmax(echo_range) for the range grid, api.py:110)."""
    t = torch.tensor([value], dtype=torch.float64)
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        t = t.to(_comm_device())
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())


def local_bin_span(local_ping_ns, e0, dt_ns, closed="left"):
    """Global indices (first, last) of the time bins this shard's pings fall in."""
    t = np.asarray(local_ping_ns, dtype=np.int64)
    t = t[t != np.iinfo(np.int64).min]
    if t.size == 0:
        return 0, -1
    if closed == "left":
        b = (t - e0) // dt_ns
    else:
        b = -((-(t - e0)) // dt_ns) - 1
    return int(b.min()), int(b.max())


def merge_straddling_bins(ssum, cnt, first_bin, last_bin, group=None):
    """Merge partial sums of time bins shared between ranks.

    ssum, cnt : (C, n_local_bins, n_rbins) raw linear sums / counts of THIS rank's local bins,
                local bin j == global bin first_bin + j.  Modified in place: after the call every
                shared bin holds the global total on its OWNER (lowest rank that has it).
    Returns ``keep``: boolean mask over local bins, False for shared bins owned by another rank.
    """
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    n_local = ssum.shape[1]
    keep = np.ones(n_local, dtype=bool)
    if world == 1 or n_local == 0 and world == 1:
        return keep
    dev = ssum.device
    # collectives run on the backend's device (GPU for RCCL, host for gloo); partials may live elsewhere
    cdev = _comm_device() if dist.get_backend(group) != "nccl" or not ssum.is_cuda else dev
    C, _, R = ssum.shape
    # 1. everyone learns every rank's (first, last) global bin ids
    ids = torch.full((world, 2), -1, dtype=torch.int64, device=cdev)
    if n_local > 0:
        ids[rank, 0], ids[rank, 1] = first_bin, last_bin
    span = torch.zeros((world, 2), dtype=torch.int64, device=cdev)
    span[rank] = ids[rank] + 1  # +1 so that "no bins" (-1) sums as 0
    dist.all_reduce(span, op=dist.ReduceOp.SUM, group=group)
    span = (span - 1).cpu().numpy()
    # 2. one all-reduce over the edge-bin partials (slot 2r = first bin of rank r, 2r+1 = last)
    buf_s = torch.zeros((2 * world, C, R), dtype=ssum.dtype, device=cdev)
    buf_c = torch.zeros((2 * world, C, R), dtype=torch.int64, device=cdev)
    if n_local > 0:
        buf_s[2 * rank] = ssum[:, 0].to(cdev)
        buf_c[2 * rank] = cnt[:, 0].to(cdev, torch.int64)
        if n_local > 1:  # a single local bin is contributed once
            buf_s[2 * rank + 1] = ssum[:, -1].to(cdev)
            buf_c[2 * rank + 1] = cnt[:, -1].to(cdev, torch.int64)
    dist.all_reduce(buf_s, op=dist.ReduceOp.SUM, group=group)
    dist.all_reduce(buf_c, op=dist.ReduceOp.SUM, group=group)
    if n_local == 0:
        return keep
    # 3. totals of my edge bins = sum over every slot carrying the same global bin id
    slot_bin = np.full(2 * world, -2, dtype=np.int64)
    for r in range(world):
        f, l = span[r]
        if f >= 0:
            slot_bin[2 * r] = f
            if l != f:
                slot_bin[2 * r + 1] = l
    for j, g in ((0, first_bin), (n_local - 1, last_bin)):
        slots = np.flatnonzero(slot_bin == g)
        owners = sorted({int(s) // 2 for s in slots})
        if len(owners) <= 1:
            continue
        idx = torch.as_tensor(slots, device=cdev)
        ssum[:, j] = buf_s.index_select(0, idx).sum(dim=0).to(dev)
        cnt[:, j] = buf_c.index_select(0, idx).sum(dim=0).to(dev, cnt.dtype)
        if owners[0] != rank:
            keep[j] = False
        if n_local == 1:
            break
    return keep
