"""Ping-sharded multi-GPU execution: one process per GPU, ``torch.distributed`` (backend "nccl" = RCCL over
xGMI on ROCm; "gloo" for the CPU tests and single-GPU dry runs).

The path shards along ``ping_time``: calibration is independent per ping; the only cross-shard
quantities are
  * scalars fixed before sharding -- EK60 tau_effective (ping 0 of the whole file), the day origin of the
    ping-time bins and the range-bin grid (all-reduce MIN / MAX of one number each, once per dataset);
  * MVBS time bins that straddle a shard edge (commongrid/utils.py:614-627 sums a bin over ALL its pings);
  * background-noise ping blocks that straddle a shard edge: the block mean comes BEFORE the minimum over range
    blocks (/root/reference/echopype/clean/api.py:402-411), so the raw (sum, count) per range block of the cut
    block are merged, not the per-shard minima.
Both exchanges have the same shape and share one mechanism, ``EdgeExchange``: a shard is a list of SEGMENTS
(resident tiles; one per rank in the simplest case), every segment contributes the raw linear (sum, count) rows
of its FIRST and LAST bin / block, one all-reduce(SUM) moves a (slots, 2, C, R) fp64 buffer (a few hundred KB:
latency-bound on xGMI, never bandwidth-bound), and every holder of a shared bin reads the total back.
Building a plan (the slot topology) is collective: two small all-reduces and host reads, once per layout -- the
product entry points below build it per call (a dataset is calibrated once); a ``ShardContext`` passed as ``shard=``
keeps plans between calls on the same layout (one scalar all-reduce per call checks that every rank still holds
its plan).  A data step is: epa_edge_pack (device) -> all_reduce -> epa_edge_gather / epa_edge_finalize_mvbs
(device) -- hand-written kernels on torch's stream, no host synchronisation, no library GEMM.
When no bin is shared (shard edges on bin edges) the exchange is skipped: no data collective at all.

Product entry points (same signatures as the single-process functions plus ``group`` / ``ping_offset`` /
``file_scalars``): ``compute_Sv``, ``compute_TS``, ``compute_MVBS``, ``compute_Sv_MVBS``, ``remove_background_noise``; and
``file_scalars`` -- the whole-file facts (first ping's pulse length, first valid ping per channel, the EK80 replica's
transmit parameters, the filter intervals' starts) a shard cannot know by itself, gathered in two or three small control
messages.  On device-resident shards the calls take the same no-wait routes as the single-process ones: one kernel for
``compute_Sv`` -> ``compute_MVBS``, two sweeps of the samples for the chain with ``remove_background_noise`` in between;
a rank whose kernel declines takes every rank to the fallback (the vote rides with the exchange plan's message).
Kernels are called through ops.
"""
import numpy as np
import torch
import torch.distributed as dist

NO_BIN = -(2**62)  # id of an absent edge (matches nothing)


def _world(group=None):
    return dist.get_world_size(group) if dist.is_initialized() else 1


def _collective(group=None):
    """Collectives run whenever a process group exists -- also at world size 1, where they are identities: the code
    path (communication buffer in HBM under RCCL, the all-reduce on it) is then the one N > 1 takes."""
    return dist.is_initialized()


def _rank(group=None):
    return dist.get_rank(group) if dist.is_initialized() else 0


def _comm_device(group=None):
    if dist.is_initialized() and dist.get_backend(group) == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


# ---- control plane: host-side agreements travel over gloo, never over the device streams ----------------------------
# A collective on the RCCL group is ordered behind the kernels already queued on the device, and reading its result
# makes the host wait for them: a handful of scalar agreements per call (time grid, layout digests, fallback decisions)
# would serialise every rank's host with its own 10-ms kernels.  Those scalars are host values to begin with, so they go
# over a gloo group of the same ranks (TCP / shared memory between the processes of one node, ~100 us each); RCCL over
# xGMI carries what lives in HBM: the cut-bin (sum, count) rows and the range maximum.
_control = {}
SEPARATE_CONTROL_GROUP = False  # test hook: a gloo control group of its own even when the data group is a gloo group


def control_group(group=None):
    """The gloo group that carries the host-side scalars of ``group`` (``group`` itself when it is a gloo group).
    Created on first use by the MEMBERS of ``group`` only (``use_local_synchronization``: ``dist.new_group`` otherwise
    has to be entered by every rank of the default group, and a sub-group's first sharded call is made by its members
    alone -- the others would never arrive).  ``register_control_group`` hands in one created elsewhere."""
    if not dist.is_initialized():
        return None
    world_pg = dist.distributed_c10d._get_default_group()
    hit = _control.get(group)
    if hit is not None and hit[0] is world_pg:
        return hit[1]
    if dist.get_backend(group) == "gloo" and not SEPARATE_CONTROL_GROUP:
        ctl = group
    else:
        ranks = dist.get_process_group_ranks(group if group is not None else world_pg)
        ctl = dist.new_group(ranks=ranks, backend="gloo", use_local_synchronization=True)
    _control[group] = (world_pg, ctl)
    return ctl


def register_control_group(group, control):
    """Use ``control`` (a gloo group of exactly the ranks of ``group``, created by the caller) for the host-side scalars
    of ``group`` instead of creating one on first use."""
    world_pg = dist.distributed_c10d._get_default_group()
    want = dist.get_process_group_ranks(group if group is not None else world_pg)
    if dist.get_backend(control) != "gloo" or dist.get_process_group_ranks(control) != want:
        raise ValueError("the control group must be a gloo group of the same ranks as the data group")
    _control[group] = (world_pg, control)


def _host_allreduce(values, op, group=None, dtype=torch.int64):
    """All-reduce of a few host scalars over the control group -> list of Python numbers (no device stream involved)."""
    t = torch.tensor(np.asarray(values).tolist() if isinstance(values, np.ndarray) else list(values), dtype=dtype)
    if _collective(group):
        dist.all_reduce(t, op=op, group=control_group(group))
    return t.tolist()


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
    lo = int(t.min()) if t.size else np.iinfo(np.int64).max
    hi = int(t.max()) if t.size else np.iinfo(np.int64).min + 1
    first, neg_last = _host_allreduce([lo, -hi], dist.ReduceOp.MIN, group)  # (one message: max = -min of the negation)
    last = -neg_last
    day = 86400 * 10**9
    origin = (first // day) * day
    e0 = origin + ((first - origin) // dt_ns) * dt_ns
    return e0, int((last - e0) // dt_ns + 1)


def global_grid(local_ping_ns, dt_ns, reach, group=None, sorted_valid=False, route=None):
    """``global_time_grid`` and ``global_max`` of a non-negative number in ONE message: all-reduce MIN of
    [first timestamp, -last timestamp, -bits(reach)] (the IEEE bit pattern of a non-negative double orders like the
    number).  Returns (first_edge, n_bins_global, global reach; NaN when no rank has one).  ``sorted_valid``: the
    caller has checked that the timestamps are sorted and hold no NaT -- the ends are the extremes (no O(P) pass).
    ``route`` (a small non-negative int, or None): the message doubles as a vote on which route the call takes -- what
    each rank could take is decided by rank-local state (is its Sv still deferred? are its pings sorted?), and the
    routes run different collectives; a fourth value comes back, the route every rank named, or 0 when they differ."""
    t = np.asarray(local_ping_ns, dtype=np.int64)
    if sorted_valid and t.size:
        lo, hi = int(t[0]), int(t[-1])
    else:
        t = t[t != np.iinfo(np.int64).min]
        lo = int(t.min()) if t.size else np.iinfo(np.int64).max
        hi = int(t.max()) if t.size else np.iinfo(np.int64).min + 1
    r = float(reach)
    rbits = int(np.float64(r).view(np.int64)) if (r == r and r >= 0.0 and r != float("inf")) else -1
    msg = [lo, -hi, -rbits] + ([int(route), -int(route)] if route is not None else [])
    got = _host_allreduce(msg, dist.ReduceOp.MIN, group)
    first, neg_last, neg_rbits = got[:3]
    last = -neg_last
    day = 86400 * 10**9
    origin = (first // day) * day
    e0 = origin + ((first - origin) // dt_ns) * dt_ns
    gr = float(np.int64(-neg_rbits).view(np.float64)) if -neg_rbits >= 0 else float("nan")
    if route is None:
        return e0, int((last - e0) // dt_ns + 1), gr
    return e0, int((last - e0) // dt_ns + 1), gr, (int(got[3]) if got[3] == -got[4] else 0)


def global_max(value, group=None):
    """all-reduce MAX of one float (e.g. nanmax(echo_range) for the range grid, api.py:110); NaN = nothing here."""
    v = float(value)
    return float(_host_allreduce([v if v == v else -np.inf], dist.ReduceOp.MAX, group, dtype=torch.float64)[0])


def global_max_device(t, group=None):
    """The same maximum for a number that still lives in HBM (a 1-element f64 device tensor a kernel is about to fill;
    NaN = nothing here): all-reduce MAX in place on the DATA group, ordered on the device behind the kernel -- the host
    does not wait; whoever reads the tensor later finds the global value.  (gloo dry runs stage it through the host.)"""
    if not _collective(group):
        return t
    from . import ops

    ops.edge_prepare_max(t)
    if _comm_device(group).type == "cuda":
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    else:
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.MAX, group=group)
        t.copy_(h)
    return t


def local_bin_span(local_ping_ns, e0, dt_ns, closed="left", sorted_valid=False):
    """Global indices (first, last) of the time bins this shard's pings fall in (``sorted_valid``: see global_grid)."""
    t = np.asarray(local_ping_ns, dtype=np.int64)
    if sorted_valid and t.size:
        t = t[[0, -1]]  # (the bin index is monotone in the timestamp)
    else:
        t = t[t != np.iinfo(np.int64).min]
    if t.size == 0:
        return 0, -1
    if closed == "left":
        b = (t - e0) // dt_ns
    else:
        b = -((-(t - e0)) // dt_ns) - 1
    return int(b.min()), int(b.max())


class EdgeExchange:
    """Totals of the bins / blocks shared between segments of a ping-sharded dataset.

    ``spans``: [(first_id, last_id)] global bin ids of THIS rank's segments, in order ((0, -1) = empty segment).
    After ``plan = EdgeExchange(spans, C, R, device)``:
      plan.shared           -- False when no bin is held by two segments anywhere (merge() is then a no-op)
      plan.edges            -- [(segment, 0 | 1, bin id, owner)] local edges that are shared; ``owner`` is True on
                               the globally first segment holding the bin (it reports the bin, the others drop it)
      totals = plan.merge(rows) -- rows[(segment, which)] = (sum (C, R), count (C, R)) raw linear partials of every
                               local edge listed in plan.edges; returns {(segment, which): (sum, count)} fp64
                               totals over all segments of all ranks.  ONE all-reduce.
      plan.merge_mvbs(rows, dst, fill) -- the same exchange, then the owners' cut bins finalised
                               (10 log10(sum / count)) straight into their rows dst[(segment, which)] of the MVBS arrays.
    A segment with a single bin contributes it once (as its first edge).

    Building a plan is collective (two small all-reduces + host reads of the topology); merge() on device rows is
    epa_edge_pack -> all_reduce -> epa_edge_gather / epa_edge_finalize_mvbs on torch's current stream: hand-written
    kernels, no host synchronisation, no library GEMM.  Under RCCL ("nccl") the communication buffer lives in HBM;
    under gloo (dry runs) it is staged through a pinned host buffer.  CPU tensors (``device="cpu"``: the slot
    bookkeeping tests of tests/test_sharding_gloo.py, which have no GPU) take an index-add on the host instead of the
    two kernels -- device rows never do.
    """

    def __init__(self, spans, C, R, device, group=None):
        self.group, self.C, self.R = group, int(C), int(R)
        self.device = torch.device(device)
        self.key = (tuple((int(f), int(l)) for f, l in spans), self.C, self.R, str(self.device))
        world, rank = _world(group), _rank(group)
        self.max_seg = int(_host_allreduce([len(spans)], dist.ReduceOp.MAX, group)[0])
        ids = np.full((world, self.max_seg, 2), NO_BIN, dtype=np.int64)
        for k, (f, l) in enumerate(spans):
            if l >= f:
                ids[rank, k, 0] = f
                if l != f:
                    ids[rank, k, 1] = l
        if _collective(group):  # every rank fills its own rows, the others are 0 after the shift (host scalars: control group)
            t = torch.zeros((world, self.max_seg, 2), dtype=torch.int64)
            t[rank] = torch.from_numpy(ids[rank] - NO_BIN)
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=control_group(group))
            ids = t.numpy() + NO_BIN
        flat = ids.reshape(-1)  # slot = (rank * max_seg + segment) * 2 + which
        self.n_slots = flat.size
        groups = {}
        for slot, b in enumerate(flat):
            if b != NO_BIN:
                groups.setdefault(int(b), []).append(slot)
        self.edges, goff, gslots = [], [0], []
        base = rank * self.max_seg * 2
        for k in range(len(spans)):
            for which in (0, 1):
                b = int(flat[base + 2 * k + which])
                if b == NO_BIN or len(groups[b]) < 2:
                    continue
                self.edges.append((k, which, b, groups[b][0] == base + 2 * k + which))
                gslots += groups[b]  # ascending slot order: the same order of additions on every holder
                goff.append(len(gslots))
        self.shared = any(len(g) > 1 for g in groups.values())
        self._slot = {(k, w): base + 2 * k + w for k, w, _, _ in self.edges}
        self._index = {(k, w): i for i, (k, w, _, _) in enumerate(self.edges)}
        self._goff_host, self._gslots_host = np.asarray(goff, dtype=np.int32), np.asarray(gslots, dtype=np.int32)
        self._buf = self._hbuf = self._goff = self._gslots = None
        if not self.shared:
            return
        self._buf = torch.zeros((self.n_slots, 2, self.C, self.R), dtype=torch.float64, device=self.device)
        if self.device.type == "cuda":
            self._goff = torch.from_numpy(self._goff_host).to(self.device)
            self._gslots = torch.from_numpy(self._gslots_host if gslots else np.zeros(1, np.int32)).to(self.device)
            if _collective(group) and _comm_device(group).type == "cpu":  # gloo dry run: staged through pinned host memory
                self._hbuf = torch.empty(self._buf.shape, dtype=torch.float64, pin_memory=True)

    @property
    def nbytes(self):
        """Bytes one all-reduce moves."""
        return 0 if self._buf is None else self._buf.numel() * 8

    def _exchange(self, rows):
        """pack -> all-reduce; leaves the reduced slots in self._buf."""
        buf = self._buf
        if self.device.type == "cuda":
            from . import ops

            ops.edge_pack(buf, [(self._slot[key], s, c) for key, (s, c) in rows.items() if key in self._slot])
            if _collective(self.group):
                if self._hbuf is None:  # RCCL: the buffer is reduced where it lies
                    dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group)
                else:
                    self._hbuf.copy_(buf)
                    dist.all_reduce(self._hbuf, op=dist.ReduceOp.SUM, group=self.group)
                    buf.copy_(self._hbuf)
            return
        buf.zero_()  # CPU tensors: bookkeeping tests
        for key, (s, c) in rows.items():
            slot = self._slot.get(key)
            if slot is not None:
                buf[slot, 0].copy_(s)
                buf[slot, 1].copy_(c)
        if _collective(self.group):
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group)

    def merge(self, rows):
        if not self.shared:
            return {}
        self._exchange(rows)
        if not self.edges:
            return {}
        if self.device.type == "cuda":
            from . import ops

            tot = ops.edge_gather(self._buf, self._goff, self._gslots, len(self.edges))
        else:
            tot = torch.stack([self._buf[self._gslots_host[a:b].tolist()].sum(0)
                               for a, b in zip(self._goff_host[:-1], self._goff_host[1:])])
        return {(k, w): (tot[i, 0], tot[i, 1]) for i, (k, w, _, _) in enumerate(self.edges)}

    def merge_mvbs(self, rows, dst, fill_value=float("nan")):
        """The exchange, then every OWNED cut bin finalised into ``dst[(segment, which)]`` ((C, R) views of the MVBS
        arrays, device): two kernel launches around the all-reduce for any number of bins up to 16."""
        if not self.shared:
            return
        if self.device.type != "cuda":  # CPU tensors (bookkeeping tests): totals, then the same formula with torch
            for (k, w), (s, c) in self.merge(rows).items():
                if self.edges[self._index[(k, w)]][3]:
                    d = dst[(k, w)]
                    d.copy_(torch.where(c > 0, 10.0 * torch.log10(s / c.clamp_min(1.0)),
                                        torch.full_like(s, fill_value)).to(d.dtype))
            return
        self._exchange(rows)
        from . import ops

        ops.edge_finalize_mvbs(self._buf, self._goff, self._gslots,
                               [(self._index[(k, w)], dst[(k, w)]) for k, w, _, owner in self.edges if owner], fill_value)


class ShardContext:
    """Keeps exchange plans between calls on the same layout (a plan costs two small all-reduces and host reads; a
    pipeline that calls the sharded functions once per file of a survey, or the bench once per step, builds it once).
    A plan depends on EVERY rank's spans (slot groups, ``shared``, the owner flags), so the cache is keyed on the
    global layout: ``plan()`` is collective -- one small all-reduce hands every rank the digests of all ranks' local
    keys, and the tuple of them is the cache key.  Every rank therefore hits or misses together (a rank whose layout
    changed makes all ranks rebuild; a rank that returns to an earlier layout finds a plan only if all the others are
    on the layout they had then, too), and no rank can run a different collective sequence from the others."""

    def __init__(self, group=None):
        self.group = group
        self._plans = {}

    def _global_key(self, key, flag=False):
        """(digests of every rank's layout, True if ``flag`` is set on any rank) -- ONE control message: the vote rides
        in an extra slot of the gather."""
        import hashlib

        h = int.from_bytes(hashlib.blake2b(repr(key).encode(), digest_size=8).digest(), "little", signed=True)
        if not _collective(self.group):
            return (h,), bool(flag)
        w = _world(self.group)
        t = torch.zeros(w + 1, dtype=torch.int64)
        t[_rank(self.group)] = h
        t[w] = 1 if flag else 0
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=control_group(self.group))  # (one non-zero term per element: a gather)
        vals = t.tolist()
        return tuple(int(v) for v in vals[:w]), vals[w] > 0

    def plan(self, spans, C, R, device, declined=None):
        """The exchange plan of this layout (kept between calls).  ``declined`` (bool): the call doubles as the vote on a
        fallback -- returns None when any rank declined (every rank then takes the fallback, nobody builds a plan)."""
        key, any_declined = self._global_key((tuple((int(f), int(l)) for f, l in spans), int(C), int(R),
                                              str(torch.device(device))), bool(declined))
        if declined is not None and any_declined:
            return None
        if key not in self._plans:
            self._plans[key] = EdgeExchange(spans, C, R, device, self.group)
        return self._plans[key]

    def agree(self, flag):
        """True on every rank if ``flag`` is true on any (all-reduce MAX of one int): decisions that change the
        sequence of collectives that follows (a fallback after a rank-local error) are taken together."""
        return bool(_host_allreduce([1 if flag else 0], dist.ReduceOp.MAX, self.group)[0])


# ---- MVBS time bins ----------------------------------------------------------------------------------------------

def mvbs_edge_rows(ssum, cnt):
    """The rows EdgeExchange wants from one segment's raw partials (C, n_bins, R)."""
    out = {0: (ssum[:, 0], cnt[:, 0])}
    if ssum.shape[1] > 1:
        out[1] = (ssum[:, -1], cnt[:, -1])
    return out


def merge_straddling_bins(ssum, cnt, first_bin, last_bin, group=None, shard=None):
    """One-segment convenience form: merge the partial sums of time bins shared between ranks.
    ssum, cnt : (C, n_local_bins, R) raw linear sums / counts, local bin j == global bin first_bin + j; modified
    in place so that every shared bin holds the global total.  Returns ``keep`` (bool per local bin): False for
    shared bins reported by a lower rank."""
    n_local = ssum.shape[1]
    keep = np.ones(n_local, dtype=bool)
    spans = [(first_bin, last_bin) if n_local else (0, -1)]
    plan = (shard.plan if shard is not None else lambda *a: EdgeExchange(*a, group))(spans, ssum.shape[0], ssum.shape[2],
                                                                                   ssum.device)
    rows = {(0, w): r for w, r in mvbs_edge_rows(ssum, cnt).items()} if n_local else {}
    for (k, w), (s, c) in plan.merge(rows).items():
        j = 0 if w == 0 else n_local - 1
        ssum[:, j] = s.to(ssum.dtype)
        cnt[:, j] = c.to(cnt.dtype)
    for k, w, _, owner in plan.edges:
        if not owner:
            keep[0 if w == 0 else n_local - 1] = False
    return keep


# ---- background-noise ping blocks ---------------------------------------------------------------------------------

def noise_block_span(ping_offset, P, ping_num):
    """Global ids (first, last) of the ping blocks a shard of P pings starting at global ping ``ping_offset`` touches."""
    return (ping_offset // ping_num, (ping_offset + P - 1) // ping_num) if P > 0 else (0, -1)


def merge_noise_edges(noise, edge_sum, edge_cnt, ping_offset, P, ping_num, noise_max=float("nan"), group=None,
                      finalize=None, shard=None):
    """Replace the noise of the shard's first / last ping block by the value over the WHOLE block when a shard edge
    cuts it (clean/api.py:402-411: mean over the block, then dB, then min over range blocks).
    noise (C, n_blocks) f64, edge_sum / edge_cnt (2, C, Sb) from ops.noise_estimate(want_edges=True); in place.
    ``finalize(sum (rows, Sb), cnt (rows, Sb)) -> (rows,)``: epa_noise_finalize by default."""
    C, nb = noise.shape
    spans = [noise_block_span(ping_offset, P, ping_num)]
    plan = shard.plan(spans, C, edge_sum.shape[2], noise.device) if shard is not None else \
        EdgeExchange(spans, C, edge_sum.shape[2], noise.device, group)
    rows = {(0, 0): (edge_sum[0], edge_cnt[0])}
    if nb > 1:
        rows[(0, 1)] = (edge_sum[1], edge_cnt[1])
    tot = plan.merge(rows)
    if not tot:
        return noise
    if finalize is None:
        from . import ops

        finalize = lambda s, c: ops.noise_finalize(s.contiguous(), c.contiguous(), noise_max)  # noqa: E731
    for (k, w), (s, c) in tot.items():
        noise[:, 0 if w == 0 else nb - 1] = finalize(s, c)
    return noise


def remove_background_noise(ds_Sv, ping_num, range_sample_num, background_noise_max=None, SNR_threshold="3.0dB", *,
                            ping_offset, group=None, shard=None):
    """clean.remove_background_noise on THIS rank's ping shard of a longer dataset: ``ping_offset`` = global index
    of the shard's first ping.  Adds Sv_noise / Sv_corrected to ``ds_Sv`` exactly as the single-process call on
    the whole dataset would for these pings."""
    from .clean import api as clean_api

    return clean_api.remove_background_noise(ds_Sv, ping_num, range_sample_num, background_noise_max, SNR_threshold,
                                             _shard=(int(ping_offset), group, shard))


# ---- compute_MVBS / compute_Sv_MVBS on a shard -----------------------------------------------------------------------

class MVBSShard(ShardContext):
    """Hooks the single-process compute_MVBS calls when it runs on one rank's shard (commongrid/api.py); a
    ShardContext, so an instance handed to several calls keeps its exchange plans."""

    def time_grid(self, ns, dt, closed):
        e0, n_glob = global_time_grid(ns, dt, self.group)
        first, last = local_bin_span(ns, e0, dt, closed)
        return e0, n_glob, first, last

    def grid(self, ns, dt, closed, reach, sorted_valid=False, route=None):
        """time_grid + range_max(reach) in one control message (reach >= 0 or NaN); with ``route`` also the vote on
        the route of the call (global_grid): a sixth value, the agreed route (0: the ranks differ)."""
        got = global_grid(ns, dt, reach, self.group, sorted_valid=sorted_valid, route=route)
        e0, n_glob, gr = got[:3]
        first, last = local_bin_span(ns, e0, dt, closed, sorted_valid=sorted_valid)
        return (e0, n_glob, first, last, gr) + ((got[3],) if route is not None else ())

    def range_max(self, hi):
        return global_max(hi, self.group)

    def range_max_device(self, t):
        return global_max_device(t, self.group)

    def finish(self, res, first_bin, last_bin, fill_value, shape=None, device=None):
        """Merged, finalised MVBS of the bins this rank reports: (tensor (C, n_kept, R), index of the first kept
        local bin).  With ``shape`` = (C, local bins, R) the call is also the vote on a fallback: ``res`` None = this
        rank's kernel declined; returns None on EVERY rank if any declined (one control message for plan and vote)."""
        if shape is not None:
            C, n_local, R = shape
            plan = self.plan([(first_bin, last_bin) if n_local else (0, -1)], C, R, device, declined=res is None)
            if plan is None:
                return None
            ssum, cnt, mv = res["sum"], res["cnt"], res["MVBS"]
        else:
            ssum, cnt, mv = res["sum"], res["cnt"], res["MVBS"]
            n_local = mv.shape[1]
            plan = self.plan([(first_bin, last_bin) if n_local else (0, -1)], mv.shape[0], mv.shape[2], mv.device)
        rows = {(0, w): r for w, r in mvbs_edge_rows(ssum, cnt).items()} if n_local else {}
        plan.merge_mvbs(rows, {(0, 0): mv[:, 0], (0, 1): mv[:, n_local - 1]} if n_local else {}, fill_value)
        lo, hi = 0, n_local
        for k, w, _, owner in plan.edges:
            if owner:
                continue
            if w == 0:
                lo = 1
            else:
                hi = n_local - 1
        return mv[:, lo:max(lo, hi)], lo


def _context(group, shard):
    if shard is None:
        return MVBSShard(group)
    if not isinstance(shard, MVBSShard):
        raise TypeError("shard= takes a sharding.MVBSShard (a ShardContext with the compute_MVBS hooks)")
    return shard


def compute_MVBS(ds_Sv, range_var="echo_range", range_bin="20m", ping_time_bin="20s", skipna=True, fill_value=np.nan,
                 closed="left", range_var_max=None, *, group=None, shard=None):
    """commongrid.compute_MVBS on THIS rank's ping shard.  The time grid (day origin) and the range grid are those
    of the whole dataset; a time bin cut by a shard edge is summed over all its pings (one small all-reduce) and
    reported by the lowest rank holding it.  Returns the MVBS dataset of the bins this rank reports; concatenating
    the ranks' results along ``ping_time`` gives the single-process answer."""
    from .commongrid import api as cg_api

    return cg_api.compute_MVBS(ds_Sv, range_var, range_bin, ping_time_bin, skipna=skipna, fill_value=fill_value,
                               closed=closed, range_var_max=range_var_max, _shard=_context(group, shard))


def compute_Sv_MVBS(echodata, *, tau_effective_first_ping=None, file_scalars=None, group=None, shard=None, **kw):
    """fused.compute_Sv_MVBS on THIS rank's ping shard (see compute_MVBS above for the grid and the shared bins).
    EK60: tau_effective is ping 0 of the WHOLE file (calibrate_ek.py:154-162) -- pass the (channel,) values of the
    global first ping as ``tau_effective_first_ping`` on every rank but the first (None = this shard's own ping 0), or
    everything a shard cannot know by itself as ``file_scalars`` (``sharding.file_scalars``: EK80 complex samples,
    files with several filter intervals)."""
    from . import fused

    fs = dict(file_scalars) if file_scalars is not None else None
    if tau_effective_first_ping is not None:
        fs = dict(fs or {}, tau_nominal_first_ping=np.asarray(tau_effective_first_ping, dtype=np.float64))
    return fused.compute_Sv_MVBS(echodata, _shard=_context(group, shard), _file=fs, **kw)


# ---- whole-file scalars of a ping-sharded file ------------------------------------------------------------------------

_TX_PARAMS = ("transmit_duration_nominal", "slope", "transmit_frequency_start", "transmit_frequency_stop")
_I64_MAX = np.iinfo(np.int64).max


def _minmax_rows(a):
    """NaN-skipping (min, max) along the last axis of a 2-D array; (+inf, -inf) for a row without a number."""
    a = np.asarray(a, dtype=np.float64)
    return np.where(np.isnan(a), np.inf, a).min(axis=-1, initial=np.inf), \
        np.where(np.isnan(a), -np.inf, a).max(axis=-1, initial=-np.inf)


def file_scalars(echodata, *, waveform_mode=None, encode_mode=None, group=None):
    """The whole-file facts the reference reads off the WHOLE file and a ping shard cannot know by itself (SURVEY 8e:
    "compute these once ... and broadcast") -- collective over the control group, two or three small host messages,
    once per file; every rank gets the same dict:

      tau_nominal_first_ping   (C,)   transmit_duration_nominal at the file's FIRST ping: EK60 / GPT tau_effective
                                      (calibrate_ek.py:113-162)
      first_valid_ping_time    (C,)   int64 ns, each channel's first ping with a valid pulse length: the filter set
                                      ``assume_single_filter_time`` selects (calibrate/api.py:98-123)          [EK80]
      transmit_params          {name: (min (C,), max (C,))} of the four parameters the replica is built from -- they
                                      must not change along the file (ek80_complex.py:255-282)                 [EK80]
      interval_starts          (C, F) bool, the filter_time stamps that start a filter interval for the channel (a ping
                                      time with a valid pulse length ANYWHERE in the file; calibrate/api.py:133-160);
      interval_tau0            (C, F) the nominal pulse length at that ping;
      interval_transmit_params {name: (min (C, F), max (C, F))} over the pings of the interval   [EK80, F > 1 stamps]

    Hand the result to ``sharding.compute_Sv / compute_TS / compute_Sv_MVBS`` (``file_scalars=``).  The small groups
    (Environment, Vendor_specific, Sonar) are expected whole on every rank; only the beam group is sharded."""
    from .calibrate.calibrate_base import cp_array
    from .calibrate.calibrate_ek import retrieve_correct_beam_group
    from .echodata import BEAM1

    ek80 = echodata.sonar_model in ("EK80", "ES80", "EA640")
    beam = echodata[retrieve_correct_beam_group(echodata, "BB" if waveform_mode == "FM" else waveform_mode, encode_mode)
                    if ek80 else BEAM1]
    C, P = beam["backscatter_r"].shape[:2]
    ns = np.asarray(beam["ping_time"].values).astype("datetime64[ns]").view(np.int64)
    tau = cp_array(beam["transmit_duration_nominal"], C, P) if "transmit_duration_nominal" in beam else np.full((C, P), np.nan)
    ok = ns != np.iinfo(np.int64).min
    # message 1 (MIN, int64): the file's first ping; each channel's first ping with a valid pulse length
    first_local = int(ns[ok].min()) if ok.any() else _I64_MAX
    fv_local = [int(ns[ok & ~np.isnan(tau[c])].min()) if (ok & ~np.isnan(tau[c])).any() else _I64_MAX for c in range(C)]
    got = _host_allreduce([first_local] + fv_local, dist.ReduceOp.MIN, group)
    first, fv = int(got[0]), np.asarray(got[1:], dtype=np.int64)
    # message 2 (MIN, float64; maxima as negated minima, "nothing here" = +inf)
    held = np.flatnonzero(ns == first)
    tau_first = tau[:, held[0]] if held.size else np.full(C, np.inf)
    parts = [np.where(np.isnan(tau_first), np.inf, tau_first)]
    F, ft = 0, None
    if ek80:
        for name in _TX_PARAMS:
            lo, hi = _minmax_rows(cp_array(beam[name], C, P)) if name in beam else (np.full(C, np.inf), np.full(C, -np.inf))
            parts += [lo, -hi]
        vend = echodata["Vendor_specific"]
        if vend.sizes.get("filter_time", 1) > 1:
            ft = np.asarray(vend["filter_time"].values).astype("datetime64[ns]").view(np.int64)
            F = ft.size
            starts = np.zeros((C, F))
            tau0 = np.full((C, F), np.inf)
            for c in range(C):
                valid = ok & ~np.isnan(tau[c])
                for f in range(F):
                    hit = np.flatnonzero(valid & (ns == ft[f]))
                    if hit.size:
                        starts[c, f], tau0[c, f] = -1.0, tau[c, hit[0]]
            parts += [starts.reshape(-1), tau0.reshape(-1)]
    got = np.asarray(_host_allreduce(np.concatenate(parts), dist.ReduceOp.MIN, group, dtype=torch.float64))
    out = {"tau_nominal_first_ping": np.where(np.isinf(got[:C]), np.nan, got[:C])}
    if not ek80:
        return out
    out["first_valid_ping_time"] = fv
    o = C
    out["transmit_params"] = {}
    for name in _TX_PARAMS:
        out["transmit_params"][name] = (got[o:o + C].copy(), -got[o + C:o + 2 * C])
        o += 2 * C
    if F:
        out["interval_starts"] = got[o:o + C * F].reshape(C, F) < 0
        t0 = got[o + C * F:o + 2 * C * F].reshape(C, F)
        out["interval_tau0"] = np.where(np.isinf(t0), np.nan, t0)
        # message 3: the transmit parameters over the pings of every (channel, interval) -- an interval runs from its
        # start to 1 ns before the channel's next start (calibrate/api.py:133-160), over all the shards it crosses
        parts = []
        for name in _TX_PARAMS:
            a = cp_array(beam[name], C, P) if name in beam else np.full((C, P), np.nan)
            lo, hi = np.full((C, F), np.inf), np.full((C, F), -np.inf)
            for c in range(C):
                order = [f for f in np.argsort(ft) if out["interval_starts"][c, f]]
                for k, f in enumerate(order):
                    keep = ok & (ns >= ft[f])
                    if k + 1 < len(order):
                        keep &= ns <= ft[order[k + 1]] - 1
                    if keep.any():
                        l, h = _minmax_rows(a[c:c + 1, keep])
                        lo[c, f], hi[c, f] = l[0], h[0]
            parts += [lo.reshape(-1), -hi.reshape(-1)]
        got = np.asarray(_host_allreduce(np.concatenate(parts), dist.ReduceOp.MIN, group, dtype=torch.float64))
        out["interval_transmit_params"] = {}
        for k, name in enumerate(_TX_PARAMS):
            out["interval_transmit_params"][name] = (got[2 * k * C * F:(2 * k + 1) * C * F].reshape(C, F).copy(),
                                                     -got[(2 * k + 1) * C * F:(2 * k + 2) * C * F].reshape(C, F))
    return out


def _compute_cal_shard(cal_type, echodata, file_scalars_, group, kw):
    from .calibrate.api import _compute_cal

    if file_scalars_ is None:
        file_scalars_ = file_scalars(echodata, waveform_mode=kw.get("waveform_mode"), encode_mode=kw.get("encode_mode"),
                                     group=group)
    return _compute_cal(cal_type, echodata, _file=file_scalars_, **kw)


def compute_Sv(echodata, *, file_scalars=None, group=None, **kw):
    """calibrate.compute_Sv on THIS rank's ping shard: same keywords, same dataset for these pings as the single-process
    call on the whole file gives.  ``file_scalars``: the result of ``sharding.file_scalars`` (None: computed here --
    collective, two or three small host messages)."""
    from .xr_lite import xarray_io

    return xarray_io()(lambda ed: _compute_cal_shard("Sv", ed, file_scalars, group, kw))(echodata)


def compute_TS(echodata, *, file_scalars=None, group=None, **kw):
    """calibrate.compute_TS on THIS rank's ping shard (see compute_Sv)."""
    from .xr_lite import xarray_io

    return xarray_io()(lambda ed: _compute_cal_shard("TS", ed, file_scalars, group, kw))(echodata)
