"""File after file through the hot path with the GPU busy back to back.

The reference processes a survey as a Python loop over files (``compute_Sv`` -> ``compute_MVBS`` per file,
/root/reference/echopype/calibrate/api.py:249-345, commongrid/api.py:30-191); with dask the files' graphs overlap.  Here
a file's two calls are one 9-ms kernel launch that the host prepares in under a millisecond, and the calls do not wait for
the GPU (deferred Sv, ``DeferredDataset``) -- what is left to arrange is WHERE consecutive files' kernels run and WHEN
their results are looked at:

* consecutive items go to alternating HIP side streams, so the kernel of file k + 1 starts beside the tail of file k's (one
  bin-owning walk leaves part of the memory system idle; two launches side by side stream the same mix 10-15 % faster,
  profiles/r05_tile_streams.txt) -- streams that were SEEN to run side by side: two streams the runtime has bound to one
  hardware queue take turns (``_runs_beside``);
* the result of item k is handed out only after the next ``lag`` items have been launched: touching a deferred dataset
  (its grid size comes back from the GPU) then never stalls the queue.

    for ds_Sv, ds_MVBS in ep.pipeline.sv_mvbs(echodatas, range_bin="1m", ping_time_bin="20s"):
        ...

``run`` is the general form (any per-item function, e.g. the three calls of the noise chain, or
``sharding.compute_Sv_MVBS`` on a rank's shards: collectives are issued in item order on every rank, whatever the
stream).  No reference counterpart; an extension like ``EchoData.to_device``.
"""
import collections

import torch

from .xr_lite import DeferredDataset

__all__ = ["Pipeline", "run", "sv_mvbs"]


def _settle(result):
    """Assemble what the item left deferred (under the item's stream: the numbers it waits for come from there)."""
    for r in (result if isinstance(result, (tuple, list)) else (result,)):
        if isinstance(r, DeferredDataset):
            r._resolve()


def _runs_beside(a, b, cycles=100_000):
    """True when kernels queued on streams ``a`` and ``b`` execute at the same time.  The HIP runtime binds a stream, at
    its first use, to one of ``GPU_MAX_HW_QUEUES`` hardware queues; two streams of one queue run their kernels one after
    the other, and which streams share a queue depends on everything the process has used before (measured, round 6,
    with the runtime's default of four queues: the two-stream loop at 0.73 of the roofline on most pairs of ten streams
    and at 0.666 -- one stream's figure -- on the pairs five apart; profiles/r06_stream_pairs.txt).  Two spin kernels of
    one workgroup each: side by side they take the time of one.  Waits for the GPU (a few hundred microseconds, once
    per process and device)."""
    if not hasattr(torch.cuda, "_sleep"):  # (a private helper of torch: without it the streams are taken as they come)
        return a is not b
    home = torch.cuda.current_stream()

    def timed(streams):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(home)
        for st in streams:
            st.wait_event(e0)
            with torch.cuda.stream(st):
                torch.cuda._sleep(cycles)
        for st in streams:
            home.wait_stream(st)
        e1.record(home)
        e1.synchronize()
        return e0.elapsed_time(e1)

    timed([a, b])  # (first use binds the streams)
    one = min(timed([a]) for _ in range(2))
    both = min(timed([a, b]) for _ in range(2))
    return both < 1.5 * one


class _Streams:
    """``n`` side streams of the current device that run side by side (``_runs_beside``: a candidate that shares a
    hardware queue with a stream already chosen is set aside, up to ``_TRIES`` candidates per slot), kept per
    (device, n): a loop that calls ``run`` once per batch does not create streams -- or wait for the check -- every time."""

    _cache = {}
    _TRIES = 8
    checked = {}  # (device, n) -> every pair of the set was seen running side by side

    @classmethod
    def get(cls, n):
        dev = torch.cuda.current_stream().device
        key = (dev, n)
        if key not in cls._cache:
            chosen, ok = [], True
            for _ in range(n):
                cand = None
                for _ in range(cls._TRIES):
                    cand = torch.cuda.Stream(device=dev)
                    if all(_runs_beside(c, cand) for c in chosen):
                        break
                else:
                    ok = False  # (GPU_MAX_HW_QUEUES=1, or every queue taken: the last candidate serves)
                chosen.append(cand)
            cls._cache[key], cls.checked[key] = chosen, ok
        return cls._cache[key]


class Pipeline:
    """The loop of ``run`` taken apart, for a caller that decides item by item when to launch: ``submit(item)`` launches
    ``fn(item)`` under the next side stream and returns the results that are due now (those of the items launched
    ``lag`` items ago: zero or one), ``drain()`` returns what is still in flight.  See ``run`` for the ordering
    guarantees."""

    def __init__(self, fn, *, streams=2, lag=1, settle=True):
        n = max(0, int(streams))
        self.fn, self.lag, self.settle = fn, max(0, int(lag)), settle
        self.home = torch.cuda.current_stream()
        self.pool = _Streams.get(n) if n else [self.home]
        self.pending = collections.deque()
        self.k = 0

    def _hand_out(self):
        result, st = self.pending.popleft()
        if self.settle:
            with torch.cuda.stream(st):
                _settle(result)
        if st is not self.home:
            self.home.wait_stream(st)
        return result

    def submit(self, item):
        st = self.pool[self.k % len(self.pool)]
        self.k += 1
        if st is not self.home:
            st.wait_event(self.home.record_event())
        with torch.cuda.stream(st):
            self.pending.append((self.fn(item), st))
        out = []
        while len(self.pending) > self.lag:
            out.append(self._hand_out())
        return out

    def drain(self):
        out = []
        while self.pending:
            out.append(self._hand_out())
        return out


def run(items, fn, *, streams=2, lag=1, settle=True):
    """Yield ``fn(item)`` for every item, in order.  Item k is launched under side stream ``k % streams``; the CALLER's
    stream runs none of the items' kernels, it only orders them against the caller's own work:

    * whatever the caller (or the ``items`` iterator: an upload, ``EchoData.to_device``) queued on its stream before an
      item is launched is finished before the item's kernels start (an event, waited for on the device);
    * the result of item k is yielded after item ``k + lag`` has been launched, assembled (``settle``: deferred datasets
      are resolved under the item's stream) and with the caller's stream made to wait -- on the device -- for the item's
      stream: what the consumer launches or reads next sees finished arrays.

    ``streams=0`` is the plain loop on the caller's stream (results still ``lag`` items late)."""
    pipe = Pipeline(fn, streams=streams, lag=lag, settle=settle)
    for item in items:
        yield from pipe.submit(item)
    yield from pipe.drain()


def sv_mvbs(echodatas, *, streams=2, lag=1, cal_kwargs=None, **mvbs_kwargs):
    """``(ds_Sv, ds_MVBS)`` per EchoData: ``calibrate.compute_Sv(ed, **cal_kwargs)`` then
    ``commongrid.compute_MVBS(ds_Sv, **mvbs_kwargs)`` -- one sweep of the raw samples per file -- pipelined by ``run``."""
    from .calibrate.api import compute_Sv
    from .commongrid.api import compute_MVBS

    cal_kwargs = dict(cal_kwargs or {})

    def two_calls(ed):
        ds = compute_Sv(ed, **cal_kwargs)
        return ds, compute_MVBS(ds, **mvbs_kwargs)

    return run(echodatas, two_calls, streams=streams, lag=lag)
