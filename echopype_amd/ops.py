"""Tensor-level wrappers of the C ABI: device buffers are torch CUDA tensors (plumbing only --
allocation, streams, RCCL), every computation is a call into libechopype_amd.so.

All functions run on the tensors' device and torch's current stream and return torch tensors.
"""
import ctypes
import threading

import numpy as np
import torch

from . import _lib
from ._lib import call

_DT = {torch.float32: _lib.F32, torch.float64: _lib.F64}


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise ValueError("echopype_amd kernels need device (HBM) buffers; got a CPU tensor")
    if not t.is_contiguous():
        raise ValueError("echopype_amd kernels need C-contiguous buffers")
    return ctypes.c_void_p(t.data_ptr())


def torch_dtype(dtype):
    if isinstance(dtype, torch.dtype):
        td = dtype
    else:
        td = {"float32": torch.float32, "float64": torch.float64}.get(np.dtype(dtype).name)
    if td not in _DT:
        raise ValueError(f"dtype must be float32 or float64, got {dtype!r}")
    return td


class _Uploader:
    """Host -> HBM copies that never make the HOST wait for the GPU.

    ``tensor.to(device)`` from pageable memory is a blocking copy on torch's current stream: it returns only after every
    kernel queued before it has finished -- a dozen small parameter uploads per API call then serialise the host with
    the previous call's 10-ms kernel.  Here the array is staged in a pinned buffer (a pool: hipHostMalloc is slow, the
    buffers are re-used once the copy that read them has completed), copied on a dedicated upload stream
    (``non_blocking``), and the CURRENT stream is made to wait for that copy ON THE DEVICE (an event): the host moves on
    at once, kernels launched afterwards see the data.  One instance per device."""

    _MAX_POOL = 64
    _MAX_BYTES = 256 << 20  # larger arrays (bulk sample data) take the plain blocking copy: staging them twice costs more

    def __init__(self, device):
        self.device = device
        self.stream = torch.cuda.Stream(device=device)
        self.pool = []  # [pinned uint8 tensor, event of the last copy that read it]
        self.lock = threading.Lock()  # picking a staging buffer and marking it busy is one step, also across threads

    def _staging(self, nbytes):
        best = None
        for slot in self.pool:
            buf, ev = slot
            if buf.numel() >= nbytes and (ev is None or ev.query()) and (best is None or buf.numel() < best[0].numel()):
                best = slot
        if best is None:
            if len(self.pool) >= self._MAX_POOL:  # drop the smallest idle buffer
                idle = [s_ for s_ in self.pool if s_[1] is None or s_[1].query()]
                if idle:
                    self.pool.remove(min(idle, key=lambda s_: s_[0].numel()))
            best = [torch.empty(max(4096, 1 << (int(nbytes) - 1).bit_length()), dtype=torch.uint8, pin_memory=True), None]
            self.pool.append(best)
        return best

    def upload(self, a):
        """C-contiguous numpy array -> device tensor of the same shape / dtype, ordered before whatever the current
        stream is given next."""
        nbytes = a.nbytes
        if nbytes == 0 or nbytes > self._MAX_BYTES:
            w = a if a.flags.writeable else a.copy()
            return torch.from_numpy(w).to(self.device)
        tdt = torch.from_numpy(np.empty(0, dtype=a.dtype)).dtype
        cur = torch.cuda.current_stream(self.device)
        with self.lock:
            slot = self._staging(nbytes)
            staged = slot[0][:nbytes]
            staged.numpy().view(np.uint8)[:] = a.reshape(-1).view(np.uint8)
            with torch.cuda.stream(self.stream):
                out = torch.empty(a.shape, dtype=tdt, device=self.device)
                out.view(-1).view(torch.uint8).copy_(staged, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(self.stream)
            slot[1] = ev
        cur.wait_event(ev)
        out.record_stream(cur)  # (allocated on the upload stream, used on the current one)
        out._epa_ready = ev  # (a holder that hands the tensor to ANOTHER stream later makes that stream wait for it)
        return out


class HostFuture:
    """A few numbers on their way from HBM to the host (``fetch_async``).  ``.cpu()`` waits for THAT copy only -- not
    for whatever was queued on the compute stream after it -- and returns the host tensor (so ``fut.cpu().tolist()``
    reads like the blocking ``tensor.cpu().tolist()`` it replaces)."""

    __slots__ = ("_slot", "_host", "_event", "_value")

    def __init__(self, slot, host, event):
        self._slot, self._host, self._event, self._value = slot, host, event, None

    def cpu(self):
        if self._value is None:
            self._event.synchronize()
            self._value = self._host.clone()
            self._slot[1] = None  # the staging buffer may be re-used
            self._slot = self._host = None
        return self._value

    def tolist(self):
        return self.cpu().tolist()

    def item(self):
        return self.cpu().item()

    def __del__(self):
        # never read: hand the staging buffer back (a later copy into it is queued on the same download stream behind
        # this one, so the two cannot overlap)
        slot = getattr(self, "_slot", None)
        if slot is not None:
            slot[1] = None


class _Downloader:
    """HBM -> host copies of kernel by-products (range statistics, a maximum) that do not queue behind later work.

    ``tensor.cpu()`` is a copy on the CURRENT stream: issued after the next file's 10-ms kernel has been launched it
    completes only when that kernel has.  Here an event marks the point of the current stream where the numbers are
    final, a dedicated download stream waits for that event and copies into a pinned buffer, and the reader waits for
    the copy's own event.  One instance per device; buffers come from a small pinned pool."""

    def __init__(self, device):
        self.device = device
        self.stream = torch.cuda.Stream(device=device)
        self.pool = []  # [pinned uint8 tensor, busy marker]
        self.lock = threading.Lock()

    def fetch(self, t):
        t = t.contiguous()
        nbytes = t.numel() * t.element_size()
        with self.lock:
            slot = next((s_ for s_ in self.pool if s_[1] is None and s_[0].numel() >= nbytes), None)
            if slot is None:
                slot = [torch.empty(max(256, 1 << (int(nbytes) - 1).bit_length()), dtype=torch.uint8, pin_memory=True), None]
                self.pool.append(slot)
            slot[1] = True
        host = slot[0][:nbytes].view(t.dtype).view(t.shape)
        cur = torch.cuda.current_stream(self.device)
        ready = torch.cuda.Event()
        ready.record(cur)
        with torch.cuda.stream(self.stream):
            self.stream.wait_event(ready)
            host.copy_(t, non_blocking=True)
            done = torch.cuda.Event()
            done.record(self.stream)
        t.record_stream(self.stream)
        return HostFuture(slot, host, done)


_uploaders = {}
_downloaders = {}


def fetch_async(t):
    """Start copying a (small) device tensor to the host NOW, on a side stream, ordered after everything queued on the
    current stream so far; returns a :class:`HostFuture`."""
    dl = _downloaders.get(t.device)
    if dl is None:
        dl = _downloaders[t.device] = _Downloader(t.device)
    return dl.fetch(t)



def _uploader(dev):
    up = _uploaders.get(dev)
    if up is None:
        up = _uploaders[dev] = _Uploader(dev)
    return up


def to_device(a, dtype=None, device=None):
    """numpy / torch -> contiguous CUDA tensor.  Host arrays go up through :class:`_Uploader` (pinned staging, upload
    stream, a device-side wait): the call does not synchronise the host with the kernels already queued."""
    dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
    dev = torch.device(dev)
    if dev.type == "cuda" and dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    if isinstance(a, torch.Tensor):
        t = a
        if dtype is not None and t.dtype != dtype:
            t = t.to(dtype)
        if not t.is_cuda:
            t = _uploader(dev).upload(t.contiguous().numpy()) if dev.type == "cuda" else t.to(dev)
        return t.contiguous()
    a = np.ascontiguousarray(a)
    if dtype is not None:
        want = np.dtype(str(dtype).replace("torch.", ""))
        if a.dtype != want:
            a = a.astype(want)
    if a.dtype.kind == "M":  # datetime64 -> its int64 representation
        a = a.view(np.int64)
    if dev.type != "cuda":
        return torch.from_numpy(a if a.flags.writeable else a.copy()).to(dev)
    return _uploader(dev).upload(a)


# ---- what does not change from call to call stays in HBM ----------------------------------------------------------------
_SMALL = {}        # (device, dtype, shape, bytes) -> tensor: per-channel tables and vectors (<= 2 KiB), the same for every
_SMALL_MAX = 256   # file of a survey; read-only by contract (kernel inputs)


def to_device_small(a, dtype=None, device=None):
    """``to_device`` for a tiny parameter array (a (C,) vector, a (C, K) table): the device copy is kept, keyed on the
    CONTENT, and handed out again -- a call then uploads nothing for parameters it shares with the previous call (six
    staging copies and events per compute_Sv, a quarter of its host time).  Larger arrays take ``to_device``.  The
    tensors are shared: nobody writes to them."""
    a = np.ascontiguousarray(a)
    if dtype is not None:
        want = np.dtype(str(dtype).replace("torch.", ""))
        if a.dtype != want:
            a = a.astype(want)
    if a.nbytes > 2048 or a.nbytes == 0:
        return to_device(a, device=device)
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    if dev.type == "cuda" and dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    key = (dev, a.dtype.str, a.shape, a.tobytes())
    ent = _SMALL.get(key)
    if ent is None:
        if len(_SMALL) >= _SMALL_MAX:
            _SMALL.clear()
        t = to_device(a, device=dev)
        ev = getattr(t, "_epa_ready", None)
        if ev is not None:
            ev.synchronize()  # (once per distinct content: afterwards the copy is there for every stream)
        _SMALL[key] = ent = t
    return ent


_PING_NS = {}  # id(ndarray) -> (weakref to it, int64 view, sorted and valid?, {device: tensor})


def ping_time_facts(ping_time, want_device=True):
    """(int64 ns view, sorted without NaT?, int64 device tensor | None) of a datetime64[ns] ping_time coordinate.  For an array
    that cannot be written to (``EchoData.to_device`` marks the beam groups' ping_time so -- resident data, like the
    samples) the three are kept with the ARRAY OBJECT: the O(P) look at the time stamps and their 8 P-byte upload are
    then paid once per file, not once per call.  A writeable array is looked at and uploaded every time."""
    import weakref

    pt = np.asarray(ping_time)
    if pt.dtype != np.dtype("datetime64[ns]"):
        pt = pt.astype("datetime64[ns]")
    dev = torch.device("cuda", torch.cuda.current_device())
    ent = _PING_NS.get(id(pt)) if not pt.flags.writeable else None
    if ent is not None and ent[0]() is pt:
        _, ns, ok, tens = ent
    else:
        ns = pt.view(np.int64)
        # unsorted, or NaT (INT64_MIN: the smallest value, so in a sorted array it could only be the first)
        ok = bool(ns.size) and not bool(np.any(ns[1:] < ns[:-1])) and ns[0] != np.iinfo(np.int64).min
        tens = {}
        if not pt.flags.writeable:
            key = id(pt)
            _PING_NS[key] = (weakref.ref(pt, lambda _r, k=key: _PING_NS.pop(k, None)), ns, ok, tens)
    if not want_device:
        return ns, ok, None
    t = tens.get(dev)
    if t is None:
        t = to_device(ns, device=dev)
        tens[dev] = t
    else:  # (uploaded under another call's stream: this one waits for that copy on the device)
        ev = getattr(t, "_epa_ready", None)
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)
    return ns, ok, t


def _mode_of(t, C, P):
    n = t.numel()
    if n == 1:
        return _lib.PM_SCALAR
    if n == C and t.dim() <= 1:
        return _lib.PM_CHANNEL
    if n == C * P:
        return _lib.PM_CHANNEL_PING
    raise ValueError(f"parameter of shape {tuple(t.shape)} is not scalar, (C,) or (C,P) with C={C}, P={P}")


def _tau_mode(tau_eff, C, P):
    """tau_effective: (C,) per channel, or (C, P) -- an EK80 file with several filter_time intervals."""
    if tuple(tau_eff.shape) == (C,):
        return _lib.PM_CHANNEL
    if tuple(tau_eff.shape) == (C, P):
        return _lib.PM_CHANNEL_PING
    raise ValueError(f"tau_eff of shape {tuple(tau_eff.shape)}: expected ({C},) or ({C}, {P})")


def power_coef_ek(sample_interval, tau_nominal, transmit_power, sound_speed, absorption, gain, sa,
                  psi, f_nominal, tau_eff, *, sonar="EK60", cal_type="Sv", pulse_length=None,
                  gain_is_table=False, sa_is_table=False, gpt=None):
    """K0 -> coef (C, P, NCOEF) f64.  All inputs f64 CUDA tensors; ``psi`` scalar, (C,) or (C, P)."""
    C, P = sample_interval.shape
    coef = torch.empty((C, P, _lib.NCOEF), dtype=torch.float64, device=sample_interval.device)
    K = 0 if pulse_length is None else pulse_length.shape[1]
    gmode = _lib.PM_PULSE_TABLE if gain_is_table else _mode_of(gain, C, P)
    smode = _lib.PM_PULSE_TABLE if sa_is_table else _mode_of(sa, C, P)
    call("epa_power_coef_ek", C, P, _p(sample_interval), _p(tau_nominal), _p(transmit_power),
         _p(sound_speed), _mode_of(sound_speed, C, P), _p(absorption), _mode_of(absorption, C, P),
         _p(gain), gmode, _p(sa), smode, _p(pulse_length), K, _p(psi), _mode_of(psi, C, P), _p(f_nominal), _p(tau_eff),
         _tau_mode(tau_eff, C, P), _p(gpt), _lib.SONAR_EK60 if sonar in ("EK60", "ES70") else _lib.SONAR_EK80,
         _lib.CAL_SV if cal_type == "Sv" else _lib.CAL_TS, _p(coef), _stream())
    return coef


def pulse_table_lookup(tau_nominal, pulse_length, table):
    """(C, P) f64: table[c, argmin_k |tau[c,p] - pulse_length[c,k]|] (get_vend_cal_params_power on the device)."""
    C, P = tau_nominal.shape
    out = torch.empty((C, P), dtype=torch.float64, device=tau_nominal.device)
    call("epa_pulse_table_lookup", _p(tau_nominal), _p(pulse_length), _p(table), C, P, pulse_length.shape[1], _p(out),
         _stream())
    return out


def sv_power_vectorized(raw, S, dtype):
    """True when K1 takes its 16-byte path on these buffers (needed for range statistics without the range array)."""
    return S % (2 if dtype == torch.float64 else 4) == 0 and raw.data_ptr() % 16 == 0


def sv_power(raw, coef, *, cal_type="Sv", flags=_lib.FLAG_GUARD_POS | _lib.FLAG_MASK_RANGE,
             dtype=torch.float64, want_range=True, out=None, range_out=None, want_range_stats=False):
    """K1 -> (Sv|TS, echo_range|None), both (C,P,S) of dtype [, f64 device tensor {nanmin, nanmax, NaN count} of
    the echo_range, a by-product of the same pass].  ``want_range=False`` with ``want_range_stats=True``: the
    statistics of the echo_range that would be written, without the array (range_power writes it later if needed)."""
    C, P, S = raw.shape
    if raw.dtype != torch.float32:
        raise ValueError("raw power samples must be float32 (convert/parse_base.py:302)")
    if out is None:
        out = torch.empty((C, P, S), dtype=dtype, device=raw.device)
    if want_range and range_out is None:
        range_out = torch.empty((C, P, S), dtype=dtype, device=raw.device)
    ct = _lib.CAL_SV if cal_type == "Sv" else _lib.CAL_TS
    if want_range_stats:
        ws = torch.empty(3 * (S // 256 + 8192 + 1024), dtype=torch.float64, device=raw.device)
        stats = torch.empty(3, dtype=torch.float64, device=raw.device)
        call("epa_sv_power_stats", _p(raw), _p(coef), C, P, S, ct, flags, _p(out),
             _p(range_out) if want_range else None, _DT[out.dtype], _p(ws), _p(stats), _stream())
        return out, (range_out if want_range else None), stats
    call("epa_sv_power", _p(raw), _p(coef), C, P, S, ct, flags, _p(out), _p(range_out) if want_range else None,
         _DT[out.dtype], _stream())
    return out, (range_out if want_range else None)


def range_power(raw, coef, *, flags=_lib.FLAG_MASK_RANGE, dtype=torch.float64):
    """echo_range (C,P,S) of the power-sample coefficient rows alone (what K1 writes as range_out)."""
    C, P, S = raw.shape
    out = torch.empty((C, P, S), dtype=dtype, device=raw.device)
    call("epa_range_power", _p(raw), _p(coef), C, P, S, flags, _p(out), _DT[dtype], _stream())
    return out


def time_bin_offsets(ping_time_ns, t0, dt, n_bins, closed="left"):
    """CSR offsets (n_bins+1,) int32 of sorted int64-ns ping times against uniform edges."""
    P = ping_time_ns.numel()
    out = torch.empty(n_bins + 1, dtype=torch.int32, device=ping_time_ns.device)
    call("epa_time_bin_offsets", _p(ping_time_ns), P, int(t0), int(dt), int(n_bins),
         _lib.BIN_CLOSED_RIGHT if closed == "right" else 0, _p(out), _stream())
    return out


def _bin_flags(skipna, closed):
    return (_lib.BIN_SKIPNA if skipna else 0) | (_lib.BIN_CLOSED_RIGHT if closed == "right" else 0)


def reduce_needs_workspace(C, n_tbins, n_rbins, dtype):
    """Mirror of block_reduce.hip::make_plan: the two-stage path (few ping bins, or a range grid
    too large for LDS) needs caller-provided sum/count workspaces."""
    esz = 8 if dtype == torch.float64 else 4
    lds = ((n_rbins * esz + 15) & ~15) + 4 * n_rbins
    return C * n_tbins < 1024 or lds > 128 * 1024


def sv_mvbs_fused(raw, coef, bin_start, n_tbins, range_bin, n_rbins, *, cal_type="Sv",
                  cal_flags=_lib.FLAG_GUARD_POS | _lib.FLAG_MASK_RANGE, skipna=True, closed="left",
                  fill_value=float("nan"), dtype=torch.float64, want_sv=True, want_range=False,
                  want_partials=False, ping_perm=None, sv_out=None, range_out=None, mvbs_out=None,
                  want_range_max=False, want_range_stats=False):
    """K1+K5 -> dict(Sv, echo_range, MVBS, sum, cnt, range_max, range_stats).  ``want_range_stats``: f64 device tensor
    {nanmin, nanmax, NaN count} of the echo_range array that is NOT written (NaN count -1: the kernel that served the
    configuration leaves the maximum only)."""
    C, P, S = raw.shape
    dev = raw.device
    if want_sv and sv_out is None:
        sv_out = torch.empty((C, P, S), dtype=dtype, device=dev)
    if want_range and range_out is None:
        range_out = torch.empty((C, P, S), dtype=dtype, device=dev)
    if mvbs_out is None:
        mvbs_out = torch.empty((C, n_tbins, n_rbins), dtype=dtype, device=dev)
    ssum = cnt = None
    if want_partials or reduce_needs_workspace(C, n_tbins, n_rbins, dtype):
        ssum = torch.empty((C, n_tbins, n_rbins), dtype=dtype, device=dev)
        cnt = torch.empty((C, n_tbins, n_rbins), dtype=torch.int32, device=dev)
    rmax = torch.empty(1, dtype=torch.float64, device=dev) if want_range_max or want_range_stats else None
    rstats = torch.empty(3, dtype=torch.float64, device=dev) if want_range_stats else None
    call("epa_sv_mvbs_fused", _p(raw), _p(coef), C, P, S,
         _lib.CAL_SV if cal_type == "Sv" else _lib.CAL_TS, cal_flags, _p(bin_start), _p(ping_perm),
         int(n_tbins), float(range_bin), int(n_rbins), _bin_flags(skipna, closed), float(fill_value),
         _p(sv_out) if want_sv else None, _p(range_out) if want_range else None, _p(mvbs_out),
         _p(ssum), _p(cnt), _p(rmax), _p(rstats), _DT[dtype], _stream())
    return dict(Sv=sv_out if want_sv else None, echo_range=range_out if want_range else None,
                MVBS=mvbs_out, sum=ssum, cnt=cnt, range_max=rmax, range_stats=rstats,
                range_stats_filled=bool(_lib.lib.epa_last_range_stats_filled()) if want_range_stats else False)


def sv_mvbs_fused_depth(raw, coef, depth_scale, depth_offset, bin_start, n_tbins, range_bin, n_rbins, *, cal_type="Sv",
                        fill_value=float("nan"), dtype=torch.float64, want_sv=True, want_depth=False,
                        want_partials=False, sv_out=None, mvbs_out=None):
    """K1+K5 binned on depth = depth_offset[c,p] + depth_scale[c,p] * echo_range (add_depth fused into the pass) ->
    dict(Sv, depth, MVBS, sum, cnt, range_stats = f64 device tensor {nanmin, nanmax, NaN count} of the depth array).
    Raises EpaError (EPA_EUNSUPPORTED) for a configuration only the generic kernels serve."""
    C, P, S = raw.shape
    dev = raw.device
    if want_sv and sv_out is None:
        sv_out = torch.empty((C, P, S), dtype=dtype, device=dev)
    depth_out = torch.empty((C, P, S), dtype=dtype, device=dev) if want_depth else None
    if mvbs_out is None:
        mvbs_out = torch.empty((C, n_tbins, n_rbins), dtype=dtype, device=dev)
    ssum = cnt = None
    if want_partials:
        ssum = torch.empty((C, n_tbins, n_rbins), dtype=dtype, device=dev)
        cnt = torch.empty((C, n_tbins, n_rbins), dtype=torch.int32, device=dev)
    rstats = torch.empty(64, dtype=torch.float64, device=dev)
    call("epa_sv_mvbs_fused_depth", _p(raw), _p(coef), _p(depth_scale), _p(depth_offset), C, P, S,
         _lib.CAL_SV if cal_type == "Sv" else _lib.CAL_TS, _p(bin_start), int(n_tbins), float(range_bin), int(n_rbins),
         float(fill_value), _p(sv_out) if want_sv else None, _p(depth_out), _p(mvbs_out), _p(ssum), _p(cnt), _p(rstats),
         _DT[dtype], _stream())
    return dict(Sv=sv_out if want_sv else None, depth=depth_out, MVBS=mvbs_out, sum=ssum, cnt=cnt,
                range_stats=rstats[:3], range_stats_filled=True)


def sv_mvbs_fused_i16(raw_i16, n_valid, coef, bin_start, n_tbins, range_bin, n_rbins, *, cal_type="Sv",
                      fill_value=float("nan"), dtype=torch.float64, want_sv=True, want_partials=False,
                      sv_out=None, mvbs_out=None, want_range_max=False):
    """K1+K5 fed with int16 instrument samples + per-ping recorded lengths."""
    C, P, S = raw_i16.shape
    if raw_i16.dtype != torch.int16 or n_valid.dtype != torch.int32:
        raise ValueError("raw_i16 must be int16 and n_valid int32")
    dev = raw_i16.device
    if want_sv and sv_out is None:
        sv_out = torch.empty((C, P, S), dtype=dtype, device=dev)
    if mvbs_out is None:
        mvbs_out = torch.empty((C, n_tbins, n_rbins), dtype=dtype, device=dev)
    ssum = cnt = None
    if want_partials:
        ssum = torch.empty((C, n_tbins, n_rbins), dtype=dtype, device=dev)
        cnt = torch.empty((C, n_tbins, n_rbins), dtype=torch.int32, device=dev)
    rmax = torch.empty(1, dtype=torch.float64, device=dev) if want_range_max else None
    call("epa_sv_mvbs_fused_i16", _p(raw_i16), _p(n_valid), _p(coef), C, P, S,
         _lib.CAL_SV if cal_type == "Sv" else _lib.CAL_TS, _p(bin_start), int(n_tbins), float(range_bin),
         int(n_rbins), float(fill_value), _p(sv_out) if want_sv else None, _p(mvbs_out), _p(ssum), _p(cnt),
         _p(rmax), _DT[dtype], _stream())
    return dict(Sv=sv_out if want_sv else None, MVBS=mvbs_out, sum=ssum, cnt=cnt, range_max=rmax)


def mvbs(sv, bin_start, n_tbins, range_bin, n_rbins, *, range=None, coef=None, skipna=True,
         closed="left", fill_value=float("nan"), want_partials=False, ping_perm=None, coef_as_stored=False):
    """K5 on an existing Sv -> dict(MVBS, sum, cnt).  ``coef`` (power-sample coefficient rows) in place of ``range``:
    the range is evaluated in the kernel; ``coef_as_stored`` rounds it to Sv's dtype first, i.e. bins exactly as on
    the echo_range array K1 would have written."""
    C, P, S = sv.shape
    dev, dtype = sv.device, sv.dtype
    if range is not None and range.dtype != dtype:
        range = range.to(dtype)
    out = torch.empty((C, n_tbins, n_rbins), dtype=dtype, device=dev)
    ssum = cnt = None
    if want_partials or reduce_needs_workspace(C, n_tbins, n_rbins, dtype):
        ssum = torch.empty((C, n_tbins, n_rbins), dtype=dtype, device=dev)
        cnt = torch.empty((C, n_tbins, n_rbins), dtype=torch.int32, device=dev)
    call("epa_mvbs", _p(sv), _p(range), _p(coef), C, P, S, _p(bin_start), _p(ping_perm), int(n_tbins),
         float(range_bin), int(n_rbins), _bin_flags(skipna, closed) | (_lib.BIN_RANGE_AS_STORED if coef_as_stored else 0),
         float(fill_value), _p(out), _p(ssum), _p(cnt), _DT[dtype], _stream())
    return dict(MVBS=out, sum=ssum, cnt=cnt)


def selftest_lin_from_db(u):
    """10^(u/10) through the fused kernel's table-driven f64 routine (accuracy self-test)."""
    out = torch.empty_like(u)
    call("epa_selftest_lin_from_db", _p(u), _p(out), u.numel(), _stream())
    return out


def selftest_log10(x, inline=False):
    out = torch.empty_like(x)
    call("epa_selftest_log10_inline" if inline else "epa_selftest_log10", _p(x), _p(out), x.numel(), _stream())
    return out


def mvbs_finalize(ssum, cnt, fill_value=float("nan")):
    out = torch.empty_like(ssum)
    call("epa_mvbs_finalize", _p(ssum), _p(cnt), ssum.numel(), float(fill_value), _p(out),
         _DT[ssum.dtype], _stream())
    return out


def affine_rows(x, scale, offset):
    """out[c,p,:] = offset[c,p] + scale[c,p] * x[c,p,:]  (add_depth)."""
    C, P, S = x.shape
    out = torch.empty_like(x)
    call("epa_affine_rows", _p(x), _p(scale), _p(offset), C, P, S, _p(out), _DT[x.dtype], _stream())
    return out


def depth_rows(scale, offset, *, range=None, coef=None, mask_raw=None, shape=None, dtype=None, want_stats=True):
    """depth = offset[c,p] + scale[c,p] * echo_range -> (depth (C,P,S), f64 device tensor {nanmin, nanmax, NaN count} of
    it, or None without ``want_stats``: from the coefficient rows that is the loop-free one-piece kernel).  echo_range as
    the array ``range`` or evaluated from the coefficient rows ``coef`` (+ ``mask_raw``: NaN where the raw power sample
    is), then ``shape`` = (C, P, S) and ``dtype`` say what to produce."""
    if range is not None:
        C, P, S = range.shape
        dtype, dev = range.dtype, range.device
    else:
        C, P, S = shape
        dev = coef.device
    out = torch.empty((C, P, S), dtype=dtype, device=dev)
    ws = stats = None
    if want_stats:
        ws = torch.empty(3 * 16384, dtype=torch.float64, device=dev)  # EPA_DEPTH_ROWS_WS_DOUBLES
        stats = torch.empty(3, dtype=torch.float64, device=dev)
    call("epa_depth_rows", _p(range), _p(coef), _p(mask_raw), _p(scale), _p(offset), C, P, S, _p(out), _DT[dtype],
         _p(ws), _p(stats), _stream())
    return out, stats


def nanminmax(x, with_nan_count=False):
    """(nanmin, nanmax[, number of NaNs]) of a device tensor as Python floats (one 24-byte D2H copy)."""
    ws = torch.empty(3072, dtype=torch.float64, device=x.device)
    out = torch.empty(3, dtype=torch.float64, device=x.device)
    call("epa_nanminmax", _p(x), x.numel(), _DT[x.dtype], _p(ws), _p(out), _stream())
    lo, hi, nn = out.cpu().tolist()
    return (lo, hi, int(nn)) if with_nan_count else (lo, hi)


def mvbs_index(sv, ping_num, range_sample_num, range=None):
    """K5' -> (MVBS (C,Pb,Sb), echo_range block-min or None)."""
    C, P, S = sv.shape
    Pb, Sb = -(-P // ping_num), -(-S // range_sample_num)
    out = torch.empty((C, Pb, Sb), dtype=sv.dtype, device=sv.device)
    rmin = None
    if range is not None:
        if range.dtype != sv.dtype:
            range = range.to(sv.dtype)
        rmin = torch.empty((C, Pb, Sb), dtype=sv.dtype, device=sv.device)
    call("epa_mvbs_index", _p(sv), _p(range), C, P, S, int(ping_num), int(range_sample_num), _p(out),
         _p(rmin), _DT[sv.dtype], _stream())
    return out, rmin


def noise_estimate(sv, alpha2, ping_num, range_sample_num, *, range=None, coef=None,
                   noise_max=float("nan"), ping_phase=0, want_edges=False):
    """K6 -> noise (C, ceil((P + ping_phase) / ping_num)) f64 [, edge_sum f64 (2, C, Sb), edge_cnt int32 (2, C, Sb):
    raw (sum, count) per range block of the first / last ping block, for the cross-shard merge]."""
    C, P, S = sv.shape
    if range is not None and range.dtype != sv.dtype:
        range = range.to(sv.dtype)
    out = torch.empty((C, -(-(P + ping_phase) // ping_num)), dtype=torch.float64, device=sv.device)
    es = ec = None
    if want_edges:
        Sb = -(-S // range_sample_num)
        es = torch.zeros((2, C, Sb), dtype=torch.float64, device=sv.device)
        ec = torch.zeros((2, C, Sb), dtype=torch.int32, device=sv.device)
    call("epa_noise_estimate", _p(sv), _p(range), _p(coef), _p(alpha2), C, P, S, int(ping_num),
         int(range_sample_num), int(ping_phase), float(noise_max), _p(out), _p(es), _p(ec), _DT[sv.dtype], _stream())
    return (out, es, ec) if want_edges else out


def noise_finalize(ssum, cnt, noise_max=float("nan")):
    """Merged (sum, count) rows (rows, Sb) f64 -> noise per row (clean/api.py:402-422)."""
    rows, Sb = ssum.shape
    out = torch.empty(rows, dtype=torch.float64, device=ssum.device)
    call("epa_noise_finalize", _p(ssum), _p(cnt), rows, Sb, float(noise_max), _p(out), _stream())
    return out


def noise_apply(sv, alpha2, noise, ping_num, snr_threshold, *, range=None, coef=None, mask_raw=None,
                want_noise=True, want_corrected=True, want_minmax=False, ping_phase=0, minmax_async=False):
    """K7 -> (Sv_noise, Sv_corrected[, [min, max of Sv_noise, min, max of Sv_corrected]]).  ``coef`` + ``mask_raw``
    (the f32 power samples) in place of ``range``: the echo_range array evaluated in the kernel, NaN where the raw
    sample is.  ``minmax_async``: the four numbers as a ``HostFuture`` (the call does not wait for its kernel)."""
    C, P, S = sv.shape
    if range is not None and range.dtype != sv.dtype:
        range = range.to(sv.dtype)
    sn = torch.empty_like(sv) if want_noise else None
    sc = torch.empty_like(sv) if want_corrected else None
    mm = torch.empty(4, dtype=torch.float64, device=sv.device) if want_minmax else None
    if range is None and mask_raw is not None:
        call("epa_noise_apply_rows", _p(sv), _p(coef), _p(mask_raw), _p(alpha2), _p(noise), C, P, S, int(ping_num),
             int(ping_phase), float(snr_threshold), _p(sn), _p(sc), _p(mm), _DT[sv.dtype], _stream())
    else:
        call("epa_noise_apply", _p(sv), _p(range), _p(coef), _p(alpha2), _p(noise), C, P, S, int(ping_num),
             int(ping_phase), float(snr_threshold), _p(sn), _p(sc), _p(mm), _DT[sv.dtype], _stream())
    return (sn, sc, fetch_async(mm) if minmax_async else mm.cpu().tolist()) if want_minmax else (sn, sc)


def complex_coef_ek80(params, tau_eff, C, P, *, B, bb, cal_type="Sv", gpt=None):
    """EK80 complex-sample coefficient rows (C, P, NCCOEF) f64 on the device.  ``params``: name -> f64 device tensor of
    shape (), (C,) or (C, P) for the names of _lib.CCP (missing / None = unused by this mode)."""
    dev = tau_eff.device
    ptrs = (ctypes.c_void_p * len(_lib.CCP))()
    modes = (ctypes.c_int * len(_lib.CCP))()
    keep = []
    for k, name in enumerate(_lib.CCP):
        t = params.get(name)
        if t is None:
            ptrs[k], modes[k] = None, _lib.PM_SCALAR
            continue
        t = t.to(torch.float64).contiguous()
        keep.append(t)
        ptrs[k], modes[k] = t.data_ptr(), _mode_of(t, C, P)
    out = torch.empty((C, P, _lib.NCCOEF), dtype=torch.float64, device=dev)
    call("epa_complex_coef_ek80", C, P, ptrs, modes, _p(tau_eff), _tau_mode(tau_eff, C, P), _p(gpt), int(B), 1 if bb else 0,
         _lib.CAL_SV if cal_type == "Sv" else _lib.CAL_TS, _p(out), _stream())
    return out


def sv_complex_uses_fft(replica, max_taps, method="auto"):
    """The form sv_complex picks: LDS-FFT for replicas of 16 .. 1024 taps (or on request), direct otherwise / CW."""
    return replica is not None and (method == "fft" or (method == "auto" and 16 <= max_taps <= _lib.EK80_NFFT // 2))


def range_complex(re, ccoef, *, dtype=torch.float64):
    """echo_range (C,P,S) of complex samples alone (what the sample kernels write as range_out)."""
    C, P, S, B = re.shape
    out = torch.empty((C, P, S), dtype=dtype, device=re.device)
    call("epa_range_complex", _p(re), _DT[re.dtype], _p(ccoef), C, P, S, B, _p(out), _DT[dtype], _stream())
    return out


def power_rows_of_complex(ccoef):
    """(C, P, NCOEF) power-sample coefficient rows with the range terms of the complex-sample rows (range =
    (s * ra) * rb, nothing else set): what epa_mvbs takes in place of an echo_range array."""
    rows = torch.zeros(tuple(ccoef.shape[:2]) + (_lib.NCOEF,), dtype=torch.float64, device=ccoef.device)
    rows[..., 0] = ccoef[..., _lib.CC_RA]
    rows[..., 1] = ccoef[..., _lib.CC_RB]
    return rows


def sv_complex(re, im, ccoef, *, replica=None, replica_off=None, max_taps=0, cal_type="Sv",
               dtype=torch.float64, want_range=True, want_prx=False, method="auto", fft_dtype=None,
               want_range_stats=False, replica_id=None):
    """K3+K4 -> dict(out, echo_range, prx[, range_stats]).  ``method``: "direct" (sliding register window),
    "fft" (LDS-resident 2048-point FFT per tile) or "auto" (fft for replicas of 16 .. 1024 taps, where it is
    faster; direct otherwise and for CW).  ``fft_dtype``: arithmetic of the transform, default = ``dtype`` (float32
    output takes complex64 butterflies, as precise as that output; float64 output a complex128 transform).
    ``want_range_stats`` (fft form and CW): f64 device tensor {nanmin, nanmax, NaN count} of the echo_range as a by-product
    of the same pass -- with ``want_range=False`` of the array range_complex would write.
    ``replica_id`` (C, P) int32: one replica per (channel, filter interval) instead of one per channel
    (``replica_off`` then has one entry per replica + 1); -1 = a ping no interval covers (NaN coefficient row)."""
    C, P, S, B = re.shape
    n_rep = C if replica_id is None else int(replica_off.numel()) - 1
    if replica_id is not None and (replica is None or tuple(replica_id.shape) != (C, P) or replica_id.dtype != torch.int32):
        raise ValueError("replica_id: int32 (C, P) next to replica / replica_off")
    if re.dtype != im.dtype or re.dtype not in _DT:
        raise ValueError("backscatter_r / backscatter_i must both be float32 or float64")
    if method not in ("auto", "direct", "fft"):
        raise ValueError("method must be 'auto', 'direct' or 'fft'")
    dev = re.device
    out = torch.empty((C, P, S), dtype=dtype, device=dev)
    rng = torch.empty((C, P, S), dtype=dtype, device=dev) if want_range else None
    prx = torch.empty((C, P, S), dtype=dtype, device=dev) if want_prx else None
    cal = _lib.CAL_SV if cal_type == "Sv" else _lib.CAL_TS
    use_fft = sv_complex_uses_fft(replica, max_taps, method)
    stats = None
    if use_fft:
        W = max(C, n_rep)
        n_ws = (768 + 4 * W + 3 * W * _lib.EK80_NFFT + 3 * 1024 + 2
                + (W * P * (S // (_lib.EK80_NFFT // 2 + 1) + 1) + 63) // 64 + W * (S + 4) + 256)  # EPA_EK80_FFT_WS_DOUBLES(W, P, S)
        ws = torch.empty(n_ws, dtype=torch.float64, device=dev)
        fdt = torch_dtype(fft_dtype) if fft_dtype is not None else dtype
        if want_range_stats:
            stats = torch.empty(3, dtype=torch.float64, device=dev)
        if replica_id is None:
            call("epa_sv_complex_fft", _p(re), _p(im), _DT[re.dtype], _p(replica), _p(replica_off), int(max_taps),
                 _p(ccoef), C, P, S, B, cal, _p(out), _p(rng), _p(prx), _DT[dtype], _DT[fdt], _p(ws), _p(stats),
                 _stream())
        else:
            call("epa_sv_complex_fft_indexed", _p(re), _p(im), _DT[re.dtype], _p(replica), _p(replica_off),
                 _p(replica_id), n_rep, int(max_taps), _p(ccoef), C, P, S, B, cal, _p(out), _p(rng), _p(prx),
                 _DT[dtype], _DT[fdt], _p(ws), _p(stats), _stream())
    elif replica is None and want_range_stats:  # CW: the streaming kernel leaves the statistics as well
        ws = torch.empty(3072, dtype=torch.float64, device=dev)  # EPA_SV_COMPLEX_CW_STATS_WS_DOUBLES
        stats = torch.empty(3, dtype=torch.float64, device=dev)
        call("epa_sv_complex_cw_stats", _p(re), _p(im), _DT[re.dtype], _p(ccoef), C, P, S, B, cal, _p(out), _p(rng),
             _p(prx), _DT[dtype], _p(ws), _p(stats), _stream())
    elif replica_id is not None:
        call("epa_sv_complex_indexed", _p(re), _p(im), _DT[re.dtype], _p(replica), _p(replica_off), _p(replica_id), n_rep,
             int(max_taps), _p(ccoef), C, P, S, B, cal, _p(out), _p(rng), _p(prx), _DT[dtype], _stream())
    else:
        call("epa_sv_complex", _p(re), _p(im), _DT[re.dtype], _p(replica), _p(replica_off), int(max_taps),
             _p(ccoef), C, P, S, B, cal, _p(out), _p(rng), _p(prx), _DT[dtype], _stream())
    return dict(out=out, echo_range=rng, prx=prx, range_stats=stats)


class Timer:
    """HIP-event timer on torch's current stream (epa_timer_*)."""

    def __init__(self):
        h = ctypes.c_void_p()
        _lib.check(_lib.lib.epa_timer_create(ctypes.byref(h)), "epa_timer_create")
        self._h = h

    def start(self):
        call("epa_timer_start", self._h, _stream())

    def stop(self):
        call("epa_timer_stop", self._h, _stream())

    def elapsed_ms(self):
        ms = ctypes.c_float()
        _lib.check(_lib.lib.epa_timer_elapsed_ms(self._h, ctypes.byref(ms)), "epa_timer_elapsed_ms")
        return float(ms.value)

    def __del__(self):
        try:
            _lib.lib.epa_timer_destroy(self._h)
        except Exception:
            pass


# ---- noise masks ------------------------------------------------------------------

def range_bin_smooth(sv, *, nper=None, range=None, r0=0.0, bin=0.0, nbins=0, out=None):
    """Depth-bin smoothing of mask_impulse_noise -> up-sampled Sv like ``sv``.
    Index mode: ``nper`` samples per bin (one value for all channels of ``sv``).
    Value mode: ``range`` array + bins np.arange(r0, max + bin, bin) (``nbins`` = len(edges) - 1)."""
    C, P, S = sv.shape
    if range is not None and range.dtype != sv.dtype:
        range = range.to(sv.dtype)
    if out is None:
        out = torch.empty_like(sv)
    call("epa_range_bin_smooth", _p(sv), _p(range), C, P, S, int(nper or 0), float(r0), float(bin),
         int(nbins), _p(out), _DT[sv.dtype], _stream())
    return out


def impulse_mask(up, num_side_pings, threshold):
    """Two-sided ping comparison -> uint8 (C,P,S)."""
    C, P, S = up.shape
    out = torch.empty((C, P, S), dtype=torch.uint8, device=up.device)
    call("epa_impulse_mask", _p(up), C, P, S, int(num_side_pings), float(threshold), _p(out),
         _DT[up.dtype], _stream())
    return out


def pool_sv(sv, first_sample, num_side_pings, num_side_samples, func="nanmean", threshold=0.0,
            want_pooled=True, want_mask=True, mask_out=None):
    """Index-binning pooled Sv (reflect window) and/or the mask Sv - pooled > threshold."""
    C, P, S = sv.shape
    f = {"nanmean": _lib.POOL_NANMEAN, "nanmedian": _lib.POOL_NANMEDIAN}[func]
    pooled = torch.empty_like(sv) if want_pooled else None
    mask = (mask_out if mask_out is not None else
            torch.empty((C, P, S), dtype=torch.uint8, device=sv.device)) if want_mask else None
    ws = wc = None
    if f == _lib.POOL_NANMEAN:
        ws = torch.empty((C, P, S), dtype=torch.float64, device=sv.device)
        wc = torch.empty((C, P, S), dtype=torch.int32, device=sv.device)
    call("epa_pool_sv", _p(sv), C, P, S, int(first_sample), int(num_side_pings), int(num_side_samples),
         f, float(threshold), _p(pooled), _p(mask), _p(ws), _p(wc), _DT[sv.dtype], _stream())
    return pooled, mask


def attenuated_mask(sv, range, upper_limit, lower_limit, num_side_pings, threshold):
    C, P, S = sv.shape
    if range.dtype != sv.dtype:
        range = range.to(sv.dtype)
    out = torch.empty((C, P, S), dtype=torch.uint8, device=sv.device)
    call("epa_attenuated_mask", _p(sv), _p(range), C, P, S, float(upper_limit), float(lower_limit),
         int(num_side_pings), float(threshold), _p(out), _DT[sv.dtype], _stream())
    return out


def apply_mask(src, mask, fill_value=float("nan"), fill_array=None):
    """where(mask, src, fill); ``mask`` uint8 with src.numel() % mask.numel() == 0 (trailing-dims
    broadcast), ``fill_array`` likewise."""
    out = torch.empty_like(src)
    if fill_array is not None and fill_array.dtype != src.dtype:
        fill_array = fill_array.to(src.dtype)
    call("epa_apply_mask", _p(src), _p(mask), src.numel(), mask.numel(), float(fill_value),
         _p(fill_array), fill_array.numel() if fill_array is not None else 1, _p(out), _DT[src.dtype],
         _stream())
    return out


def apply_masks(src, masks, fill_value=float("nan"), fill_array=None, want_minmax=False):
    """where(AND of ``masks``, src, fill) in one sweep (1 to 4 uint8 masks, each tiling ``src`` over its leading dims)
    -> (out, f64 device tensor {nanmin, nanmax} of out | None)."""
    import ctypes

    if not 1 <= len(masks) <= 4:
        raise ValueError("apply_masks takes 1 to 4 masks (AND further ones with mask_and first)")
    out = torch.empty_like(src)
    if fill_array is not None and fill_array.dtype != src.dtype:
        fill_array = fill_array.to(src.dtype)
    ptrs = (ctypes.c_void_p * len(masks))(*[m.data_ptr() for m in masks])
    periods = (ctypes.c_size_t * len(masks))(*[m.numel() for m in masks])
    ws = mm = None
    if want_minmax:
        ws = torch.empty(_lib.APPLY_MASKS_WS_DOUBLES, dtype=torch.float64, device=src.device)
        mm = torch.empty(2, dtype=torch.float64, device=src.device)
    call("epa_apply_masks", _p(src), ctypes.cast(ptrs, ctypes.c_void_p), ctypes.cast(periods, ctypes.c_void_p), len(masks),
         src.numel(), float(fill_value), _p(fill_array), fill_array.numel() if fill_array is not None else 1, _p(out),
         _p(ws), _p(mm), _DT[src.dtype], _stream())
    return out, mm


def mask_and(a, b):
    """a & b, ``b`` broadcast over the leading dims of ``a``."""
    out = torch.empty_like(a)
    call("epa_mask_and", _p(a), _p(b), a.numel(), b.numel(), _p(out), _stream())
    return out


def range_step_mean(range):
    """Per-channel nanmean of the sample-to-sample range step -> numpy f64 [C]."""
    C, P, S = range.shape
    ws = torch.empty(2 * C * P, dtype=torch.float64, device=range.device)
    out = torch.empty(C, dtype=torch.float64, device=range.device)
    call("epa_range_step_mean", _p(range), C, P, S, _DT[range.dtype], _p(ws), _p(out), _stream())
    return out.cpu().numpy()


def first_not_le(x, limit):
    """Flat index of the first element not <= limit (NaN counts); x.numel() if none."""
    out = torch.empty(1, dtype=torch.int64, device=x.device)
    call("epa_first_not_le", _p(x), x.numel(), float(limit), _DT[x.dtype], _p(out), _stream())
    return int(out.item())


def range_rows_check(range):
    """-> (nvalid int32 (C,P) device tensor, number of rows violating monotonicity / NaN-tail)."""
    C, P, S = range.shape
    nvalid = torch.empty((C, P), dtype=torch.int32, device=range.device)
    bad = torch.empty(1, dtype=torch.int32, device=range.device)
    call("epa_range_rows_check", _p(range), C, P, S, _DT[range.dtype], _p(nvalid), _p(bad), _stream())
    return nvalid, int(bad.item())


def pool_sv_value(sv, range, nvalid, depth_bin, num_side_pings, exclude_above, range_min, range_max,
                  func="nanmean", threshold=0.0, want_pooled=True, want_mask=True, running_sums=True):
    """Value-window pooled Sv (pool_Sv) and/or the mask Sv - pooled > threshold.  ``running_sums`` (nanmean):
    per-row double-double running sums and interval sums in a 40 B/sample workspace instead of summing every window
    (EPA_POOL_VALUE_WS_BYTES in the header); when that does not fit next to the arrays the workspace-free kernel runs
    (same results, every window summed value by value).  nanmedian: with ``running_sums`` the window is carried from
    ping to ping (a small workspace), without it every window is taken from memory."""
    C, P, S = sv.shape
    if range.dtype != sv.dtype:
        range = range.to(sv.dtype)
    f = {"nanmean": _lib.POOL_NANMEAN, "nanmedian": _lib.POOL_NANMEDIAN}[func]
    pooled = torch.empty_like(sv) if want_pooled else None
    mask = torch.empty((C, P, S), dtype=torch.uint8, device=sv.device) if want_mask else None
    ws = None
    if running_sums and func == "nanmean":
        try:
            ws = torch.empty((C * P * S * 40 + C * S * 8 + C * 8 + C * P + 7) // 8, dtype=torch.float64, device=sv.device)
        except torch.cuda.OutOfMemoryError:
            ws = None
    elif running_sums:  # nanmedian: the per-channel sample intervals (EPA_POOL_VALUE_MEDIAN_WS_BYTES)
        ws = torch.empty(2 * C * S + 2 * C, dtype=torch.int32, device=sv.device)
    call("epa_pool_sv_value", _p(sv), _p(range), _p(nvalid), C, P, S, float(depth_bin), int(num_side_pings),
         float(exclude_above), float(range_min), float(range_max), f, float(threshold), _p(pooled),
         _p(mask), _p(ws), _DT[sv.dtype], _stream())
    return pooled, mask


# ---- NASC ---------------------------------------------------------------------------

def geodesic_steps(lat, lon):
    """WGS-84 geodesic length in metres from every ping to the next (f64 device tensors (P,) in degrees; NaN for the
    last ping and NaN positions) -- commongrid/utils.py:208-231's geopy loop as one launch."""
    P = lat.numel()
    out = torch.empty(P, dtype=torch.float64, device=lat.device)
    call("epa_geodesic_steps", _p(lat), _p(lon), P, _p(out), _stream())
    return out


def nasc(sv, depth, bin_start, n_dbins, range_bin, n_rbins, skipna=True, closed="left", want_parts=False):
    """compute_raw_NASC -> NASC (C, n_dbins, n_rbins) [, sv_mean, h_mean]."""
    C, P, S = sv.shape
    if depth.dtype != sv.dtype:
        depth = depth.to(sv.dtype)
    ws = torch.empty(C * n_dbins * n_rbins * 3, dtype=torch.float64, device=sv.device)  # 24 B / cell
    out = torch.empty((C, n_dbins, n_rbins), dtype=sv.dtype, device=sv.device)
    svm = torch.empty_like(out) if want_parts else None
    hm = torch.empty_like(out) if want_parts else None
    call("epa_nasc", _p(sv), _p(depth), C, P, S, _p(bin_start), int(n_dbins), float(range_bin), int(n_rbins),
         _bin_flags(skipna, closed), _p(ws), _p(out), _p(svm), _p(hm), _DT[sv.dtype], _stream())
    return (out, svm, hm) if want_parts else out


# ---- the whole chain in two passes ---------------------------------------------------------------------

def sv_noise_fused(raw, coef, alpha2, ping_num, range_sample_num, *, cal_type="Sv",
                   flags=_lib.FLAG_GUARD_POS | _lib.FLAG_MASK_RANGE, dtype=torch.float64,
                   noise_max=float("nan"), want_sv=True, want_range=False, want_range_max=False,
                   want_range_stats=False, ping_phase=0, want_edges=False):
    """K1+K6 -> (Sv|None, echo_range|None, noise (C, ceil((P + ping_phase)/ping_num)) f64[, nanmax(echo_range)]).
    ``want_range_stats``: the last element is instead the f64 device tensor {nanmin, nanmax, NaN count} of the echo_range
    (NaN count -1: the kernel that served the configuration leaves none).  ``ping_phase`` / ``want_edges``: a ping shard
    of a longer file, as ``noise_estimate`` -- with ``want_edges`` two more elements follow: edge_sum f64 (2, C, Sb),
    edge_cnt int32 (2, C, Sb), the raw (sum, count) rows of the shard's first / last ping block."""
    C, P, S = raw.shape
    if raw.dtype != torch.float32:
        raise ValueError("raw power samples must be float32 (convert/parse_base.py:302)")
    dev = raw.device
    sv = torch.empty((C, P, S), dtype=dtype, device=dev) if want_sv else None
    rng = torch.empty((C, P, S), dtype=dtype, device=dev) if want_range else None
    noise = torch.empty((C, -(-(P + ping_phase) // ping_num)), dtype=torch.float64, device=dev)
    rmax = torch.empty(1, dtype=torch.float64, device=dev) if want_range_max or want_range_stats else None
    rstats = torch.empty(3, dtype=torch.float64, device=dev) if want_range_stats else None
    es = ec = None
    if want_edges:
        Sb = -(-S // range_sample_num)
        es = torch.zeros((2, C, Sb), dtype=torch.float64, device=dev)
        ec = torch.zeros((2, C, Sb), dtype=torch.int32, device=dev)
    call("epa_sv_noise_fused", _p(raw), _p(coef), _p(alpha2), C, P, S,
         _lib.CAL_SV if cal_type == "Sv" else _lib.CAL_TS, flags, int(ping_num), int(range_sample_num), int(ping_phase),
         float(noise_max), _p(sv), _p(rng), _p(noise), _p(es), _p(ec), _p(rmax), _p(rstats), _DT[dtype], _stream())
    edges = (es, ec) if want_edges else ()
    if want_range_stats:
        return (sv, rng, noise, rstats) + edges
    return ((sv, rng, noise, float(rmax.item())) if want_range_max else (sv, rng, noise)) + edges


def denoise_mvbs(sv, alpha2, noise, ping_num, snr_threshold, bin_start, n_tbins, range_bin, n_rbins, *,
                 range=None, coef=None, skipna=True, closed="left", fill_value=float("nan"),
                 ping_perm=None, want_noise=False, want_corrected=True, want_partials=False):
    """K7+K5 -> dict(MVBS of the corrected Sv, Sv_noise, Sv_corrected, sum, cnt)."""
    C, P, S = sv.shape
    dev, dtype = sv.device, sv.dtype
    if range is not None and range.dtype != dtype:
        range = range.to(dtype)
    sn = torch.empty_like(sv) if want_noise else None
    sc = torch.empty_like(sv) if want_corrected else None
    out = torch.empty((C, n_tbins, n_rbins), dtype=dtype, device=dev)
    ssum = cnt = None
    if want_partials or reduce_needs_workspace(C, n_tbins, n_rbins, dtype):
        ssum = torch.empty((C, n_tbins, n_rbins), dtype=dtype, device=dev)
        cnt = torch.empty((C, n_tbins, n_rbins), dtype=torch.int32, device=dev)
    call("epa_denoise_mvbs", _p(sv), _p(range), _p(coef), _p(alpha2), _p(noise), C, P, S, int(ping_num),
         float(snr_threshold), _p(bin_start), _p(ping_perm), int(n_tbins), float(range_bin), int(n_rbins),
         _bin_flags(skipna, closed), float(fill_value), _p(sn), _p(sc), _p(out), _p(ssum), _p(cnt),
         _DT[dtype], _stream())
    return dict(MVBS=out, Sv_noise=sn, Sv_corrected=sc, sum=ssum, cnt=cnt)


def sv_denoise_mvbs(raw, coef, alpha2, noise, ping_num, snr_threshold, bin_start, n_tbins, range_bin, n_rbins,
                    *, cal_type="Sv", flags=_lib.FLAG_GUARD_POS | _lib.FLAG_MASK_RANGE, dtype=torch.float64,
                    skipna=True, closed="left", fill_value=float("nan"), ping_perm=None, want_noise=False,
                    want_corrected=True, want_range=False, want_partials=False, want_minmax=False,
                    minmax_async=False, ping_phase=0):
    """K1+K7+K5 from the raw power -> dict(MVBS of the corrected Sv, Sv_noise, Sv_corrected, echo_range, sum,
    cnt, minmax = [min, max of Sv_noise, min, max of Sv_corrected] (host floats) if asked; ``minmax_async``: a
    ``HostFuture`` of the four numbers instead -- the call then does not wait for its kernel)."""
    C, P, S = raw.shape
    dev = raw.device
    mk = lambda want: torch.empty((C, P, S), dtype=dtype, device=dev) if want else None  # noqa: E731
    sn, sc, rng = mk(want_noise), mk(want_corrected), mk(want_range)
    out = torch.empty((C, n_tbins, n_rbins), dtype=dtype, device=dev)
    ssum = cnt = None
    if want_partials or reduce_needs_workspace(C, n_tbins, n_rbins, dtype):
        ssum = torch.empty((C, n_tbins, n_rbins), dtype=dtype, device=dev)
        cnt = torch.empty((C, n_tbins, n_rbins), dtype=torch.int32, device=dev)
    mm = torch.empty(4, dtype=torch.float64, device=dev) if want_minmax else None
    call("epa_sv_denoise_mvbs", _p(raw), _p(coef), _p(alpha2), _p(noise), C, P, S,
         _lib.CAL_SV if cal_type == "Sv" else _lib.CAL_TS, flags, int(ping_num), int(ping_phase), float(snr_threshold),
         _p(bin_start), _p(ping_perm), int(n_tbins), float(range_bin), int(n_rbins), _bin_flags(skipna, closed),
         float(fill_value), _p(sn), _p(sc), _p(rng), _p(out), _p(ssum), _p(cnt), _p(mm), _DT[dtype], _stream())
    return dict(MVBS=out, Sv_noise=sn, Sv_corrected=sc, echo_range=rng, sum=ssum, cnt=cnt,
                minmax=(fetch_async(mm) if minmax_async else mm.cpu().tolist()) if want_minmax else None)


# ---- cross-shard edge exchange (pack -> all-reduce -> gather) ------------------------------------------------

def edge_pack(buf, rows, zero_first=True):
    """rows: [(slot, sum (C, R) f32/f64 view with unit inner stride, cnt (C, R) int32 view)] -> slots of ``buf``
    (n_slots, 2, C, R) f64, everything else zero."""
    n_slots, _, C, R = buf.shape
    n = len(rows)
    sums, cnts = (ctypes.c_void_p * max(n, 1))(), (ctypes.c_void_p * max(n, 1))()
    strides, slots = (ctypes.c_longlong * max(n, 1))(), (ctypes.c_int * max(n, 1))()
    dt = None
    for i, (slot, s, c) in enumerate(rows):
        if not (s.is_cuda and c.is_cuda) or tuple(s.shape) != (C, R) or tuple(c.shape) != (C, R):
            raise ValueError(f"edge_pack: row {i} must be a pair of ({C}, {R}) device tensors")
        if s.stride(1) != 1 or c.stride(1) != 1 or s.stride(0) != c.stride(0) or c.dtype != torch.int32 or s.dtype not in _DT:
            raise ValueError("edge_pack: rows need unit inner stride, equal channel strides, f32/f64 sums and int32 counts")
        if dt is not None and s.dtype != dt:
            raise ValueError("edge_pack: rows of mixed dtypes")
        dt = s.dtype
        sums[i], cnts[i], strides[i], slots[i] = s.data_ptr(), c.data_ptr(), s.stride(0), int(slot)
    call("epa_edge_pack", sums, cnts, strides, slots, n, _DT[dt if dt is not None else torch.float64], C, R, n_slots,
         1 if zero_first else 0, _p(buf), _stream())
    return buf


def edge_prepare_max(t):
    """NaN -> -inf in place on an f64 device tensor about to be all-reduced with MAX."""
    if t.dtype != torch.float64:
        raise ValueError("edge_prepare_max: f64 device tensor expected")
    call("epa_edge_prepare_max", _p(t), t.numel(), _stream())
    return t


def edge_gather(buf, group_off, group_slots, n_edges, *, typed=None):
    """Totals of the shared edges from the all-reduced buffer: (n_edges, 2, C, R) f64, or with ``typed`` (a torch
    dtype) the pair (sum (n_edges, C, R) of that dtype, count (n_edges, C, R) int32) that mvbs_finalize takes."""
    n_slots, _, C, R = buf.shape
    dev = buf.device
    if typed is None:
        tot = torch.empty((n_edges, 2, C, R), dtype=torch.float64, device=dev)
        call("epa_edge_gather", _p(buf), n_slots, _p(group_off), _p(group_slots), n_edges, C, R, _p(tot), None, None,
             _lib.F64, _stream())
        return tot
    s = torch.empty((n_edges, C, R), dtype=typed, device=dev)
    c = torch.empty((n_edges, C, R), dtype=torch.int32, device=dev)
    call("epa_edge_gather", _p(buf), n_slots, _p(group_off), _p(group_slots), n_edges, C, R, None, _p(s), _p(c),
         _DT[typed], _stream())
    return s, c


def edge_finalize_mvbs(buf, group_off, group_slots, rows, fill_value=float("nan")):
    """rows: [(edge index, dst (C, R) f32/f64 view with unit inner stride)]: the merged, finalised MVBS of those shared
    edges (10 log10(sum / count) of the all-reduced totals) written straight into the rows ``dst``."""
    n_slots, _, C, R = buf.shape
    n = len(rows)
    if n == 0:
        return
    edges, dsts, strides = (ctypes.c_int * n)(), (ctypes.c_void_p * n)(), (ctypes.c_longlong * n)()
    dt = rows[0][1].dtype
    for i, (e, d) in enumerate(rows):
        if not d.is_cuda or tuple(d.shape) != (C, R) or d.stride(1) != 1 or d.dtype != dt or dt not in _DT:
            raise ValueError(f"edge_finalize_mvbs: row {i} must be a ({C}, {R}) f32/f64 device view with unit inner stride")
        edges[i], dsts[i], strides[i] = int(e), d.data_ptr(), d.stride(0)
    call("epa_edge_finalize_mvbs", _p(buf), n_slots, _p(group_off), _p(group_slots), edges, dsts, strides, n, C, R,
         float(fill_value), _DT[dt], _stream())
