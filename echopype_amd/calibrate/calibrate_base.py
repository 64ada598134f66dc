"""Shared plumbing of the calibrators (mirrors calibrate/calibrate_base.py:10-128)."""
import abc
import logging
import os
import weakref

import numpy as np
import torch

from .. import _lib, ops
from ..xr_lite import DataArray, DeviceArray, LazyDeviceArray
from .cal_params import PulseTableParam

logger = logging.getLogger("echopype_amd.calibrate")

ECHO_DIMS = ("channel", "ping_time", "range_sample")


def cp_array(v, C, P, name="parameter"):
    """scalar / (C,) / (P,) / (C,P) [DataArray or array-like] -> contiguous float64 (C, P)."""
    dims = v.dims if isinstance(v, DataArray) else None
    a = np.asarray(getattr(v, "values", v), dtype=np.float64)
    if a.ndim == 0:
        return np.full((C, P), float(a))
    if a.ndim == 1:
        if dims == ("ping_time",) or (dims is None and a.shape[0] == P and P != C):
            return np.ascontiguousarray(np.broadcast_to(a[None, :], (C, P)))
        if a.shape[0] == C:
            return np.ascontiguousarray(np.broadcast_to(a[:, None], (C, P)))
    if a.ndim == 2:
        if dims == ("ping_time", "channel"):
            a = a.T
        if a.shape == (C, P):
            return np.ascontiguousarray(a)
        if a.shape == (C, 1):
            return np.ascontiguousarray(np.broadcast_to(a, (C, P)))
    raise ValueError(f"{name} of shape {a.shape} cannot be broadcast to (channel={C}, ping_time={P})")


class PowerSource:
    """What a deferred Sv/TS of power samples is made from (``LazyDeviceArray.source``): the raw samples, the
    coefficient rows, the kernel flags, and the lazy echo_range that travels with it."""

    __slots__ = ("raw", "coef", "flags", "cal_type", "dtype", "echo_range", "raw_version", "reach_bound")

    def __init__(self, raw, coef, flags, cal_type, dtype, echo_range, reach_bound=None):
        self.raw, self.coef, self.flags, self.cal_type, self.dtype = raw, coef, flags, cal_type, dtype
        self.echo_range, self.raw_version = echo_range, raw._version
        # an upper bound of every echo_range the rows can produce, known on the HOST (None: reduce the rows on the device)
        self.reach_bound = reach_bound

    def intact(self):
        """The raw samples have not been written to since compute_Sv looked at them."""
        return self.raw._version == self.raw_version


class CalibrateBase(abc.ABC):
    """Common constructor semantics: an ECS file overrides the user dictionaries
    (calibrate_base.py:20-47); ECS parsing itself is out of scope."""

    def __init__(self, echodata, env_params=None, cal_params=None, ecs_file=None, **kwargs):
        self.echodata = echodata
        self.sonar_type = None
        self.ecs_file = ecs_file
        self.ecs_dict = {}
        if self.ecs_file is not None:
            raise NotImplementedError(
                "ecs_file: Echoview .ecs parsing (calibrate/ecs.py) is outside the accelerated hot "
                "path; pass the parameters through env_params / cal_params instead.")
        if env_params is None:  # calibrate_base.py:35-47
            self.env_params = {}
        elif isinstance(env_params, dict):
            self.env_params = env_params
        else:
            raise ValueError("'env_params' has to be None or a dict")
        if cal_params is None:
            self.cal_params = {}
        elif isinstance(cal_params, dict):
            self.cal_params = cal_params
        else:
            raise ValueError("'cal_params' has to be None or a dict")
        self.range_meter = None
        self.dtype = ops.torch_dtype(kwargs.get("dtype", "float64"))
        self.device = kwargs.get("device")
        self.fft_dtype = kwargs.get("fft_dtype")  # EK80 BB: arithmetic of the pulse-compression transform (None = dtype)
        # a ping shard of a longer file (echopype_amd.sharding.file_scalars): the whole-file facts the reference reads
        # off the whole file -- nominal pulse length of the FILE's first ping, each channel's first valid ping, the
        # transmit parameters that must not change, the filter intervals' starts.  None: the echodata is the whole file.
        self.file_scalars = kwargs.get("file_scalars")

    @abc.abstractmethod
    def compute_echo_range(self, **kwargs):
        pass

    @abc.abstractmethod
    def _cal_power_samples(self, cal_type, **kwargs):
        pass

    @abc.abstractmethod
    def compute_Sv(self, **kwargs):
        pass

    @abc.abstractmethod
    def compute_TS(self, **kwargs):
        pass

    def _add_params_to_output(self, ds_out):
        """Every env and cal parameter rides along in the output (calibrate_base.py:83-93)."""
        for group in (self.env_params, self.cal_params):
            for key, val in group.items():
                if val is None:
                    continue
                if isinstance(val, PulseTableParam) and not val.materialized and val.on_device:
                    # still only the (C, K) table: the (C, P) array is looked up when somebody reads the variable
                    # (one small kernel less in front of every file's calibration kernel)
                    lazy = LazyDeviceArray(val.shape, torch.float64, val.on_device,
                                           (lambda v=val: v.data.tensor))
                    ds_out[key] = DataArray(lazy, val.dims, attrs=val.attrs, name=key)
                elif isinstance(val, DataArray):
                    ds_out[key] = DataArray(val.data, val.dims, attrs=val.attrs, name=key)
                elif isinstance(val, str):
                    ds_out[key] = np.asarray(val)
                else:
                    a = np.asarray(val)
                    C = ds_out.sizes.get("channel")
                    P = ds_out.sizes.get("ping_time")
                    if a.ndim == 0:
                        ds_out[key] = a
                    elif a.ndim == 1 and a.shape[0] == C:
                        ds_out[key] = (("channel",), a)
                    elif a.ndim == 1 and a.shape[0] == P:
                        ds_out[key] = (("ping_time",), a)
                    elif a.ndim == 2:
                        ds_out[key] = (("channel", "ping_time"), a)
        return ds_out

    def _check_echodata_backscatter_size(self):
        """> 2 GiB warning, text as calibrate_base.py:116-128 (asserted verbatim by the reference's
        tests/calibrate/test_calibrate.py:432-441)."""
        beam = self.echodata[getattr(self, "ed_beam_group", None) or "Sonar/Beam_group1"]
        total = beam["backscatter_r"].nbytes
        if "backscatter_i" in beam and getattr(self, "encode_mode", "power") == "complex":
            total += beam["backscatter_i"].nbytes
        if total / (1024 ** 3) > 2.0:
            logger.warning(
                "The Echodata backscatter variables are large and can cause memory issues. "
                "Consider modifying the workflow that uses compute_Sv as below: "
                "Prior to `compute_Sv` run `echodata.chunk(CHUNK_DICTIONARY) "
                "and after `compute_Sv` run `ds_Sv.to_zarr(ZARR_STORE, compute=True)`. "
                "This will ensure that the computation is lazily evaluated, "
                "with the results stored directly in a Zarr store on disk, rather then in memory."
            )

    # ---- device helpers -------------------------------------------------------------------------
    def _cp_dev(self, v, C, P, name, dtype=None):
        """A per-(channel, ping) parameter as a contiguous (C, P) device tensor: straight from HBM when it already
        lives there in that shape (EchoData.to_device), through cp_array + one upload otherwise."""
        d = v.data if isinstance(v, DataArray) else None
        if isinstance(d, DeviceArray) and tuple(v.dims) == ("channel", "ping_time") and d.shape == (C, P):
            t = d.tensor
            return (t if dtype is None or t.dtype == dtype else t.to(dtype)).contiguous()
        return self._dev(cp_array(v, C, P, name), dtype)

    def _dev(self, a, dtype=None):
        if isinstance(a, DeviceArray):
            t = a.tensor
            return t if dtype is None or t.dtype == dtype else t.to(dtype)
        a = np.asarray(getattr(a, "values", a))
        if a.nbytes <= 2048:  # (per-channel vectors and pulse-length tables: kept in HBM by content, ops.to_device_small)
            return ops.to_device_small(a, dtype=dtype, device=self.device)
        return ops.to_device(a, dtype=dtype, device=self.device)

    @staticmethod
    def defer_enabled():
        """EPA_DEFER_SV=0 turns the deferred Sv off (compute_Sv then runs its kernel before it returns)."""
        return os.environ.get("EPA_DEFER_SV", "1") != "0"

    @staticmethod
    def _wrap(t, dims, attrs=None, name=None, stats=None):
        if isinstance(t, DeviceArray):  # (a LazyDeviceArray carries its statistics already)
            return DataArray(t, dims, attrs=attrs, name=name)
        return DataArray(DeviceArray(t, stats=stats), dims, attrs=attrs, name=name)

    def _sv_power_lazy_range(self, raw, coef, cal_type, flags):
        """K1 with echo_range left lazy: Sv/TS and the {nanmin, nanmax, NaN count} of echo_range come out of the pass,
        the echo_range array (8 of its 20 B/sample) is written by ``epa_range_power`` only if somebody reads it --
        ``compute_MVBS`` bins on the coefficient rows instead (the same arithmetic, the same values).  The reference's
        echo_range of a dask-backed EchoData is just as lazy."""
        C, P, S = raw.shape
        if not ops.sv_power_vectorized(raw, S, self.dtype):
            return ops.sv_power(raw, coef, cal_type=cal_type, flags=flags, dtype=self.dtype, want_range_stats=True)
        out_t, _, stats = ops.sv_power(raw, coef, cal_type=cal_type, flags=flags, dtype=self.dtype, want_range=False,
                                       want_range_stats=True)
        return out_t, self._lazy_power_range(raw, coef, flags, stats), stats

    def _deferred_sv_power(self, raw, coef, cal_type, flags):
        """Sv/TS of power samples left to its first reader (the reference's Sv of a dask-backed EchoData is as lazy):
        ``(sv, echo_range)`` as two LazyDeviceArrays.  Whoever reads ``sv.tensor`` runs K1 (``epa_sv_power_stats``: the
        array + the echo_range statistics); ``compute_MVBS`` -- the usual next call -- recognises ``sv.source`` and
        produces the array as a by-product of its own pass over the raw samples (``epa_sv_mvbs_fused``: 12 B/sample
        for the two calls instead of 12 + 8).  Same values either way (the kernels share the arithmetic)."""
        rng = self._lazy_power_range(raw, coef, flags)
        version, dtype = raw._version, self.dtype
        src = PowerSource(raw, coef, flags, cal_type, dtype, rng, reach_bound=self._host_reach_bound(raw.shape[2]))

        def make():
            if raw._version != version:
                raise RuntimeError(f"{cal_type} was left lazy by compute_{cal_type} and backscatter_r has been "
                                   "modified in place since")
            out_t, _, stats = ops.sv_power(raw, coef, cal_type=cal_type, flags=flags, dtype=dtype, want_range=False,
                                           want_range_stats=True)
            if rng.coef_rows() is not None:  # (the echo_range array, if somebody has read it, is still untouched)
                rng.set_stats(stats)
            return out_t

        sv = LazyDeviceArray(tuple(raw.shape), dtype, raw.device, make, source=src)
        sv_ref = weakref.ref(sv)  # (no cycle sv -> source -> echo_range -> hook -> sv: a dropped dataset frees at once)

        def stats_with_sv():  # statistics asked for first: they come with the Sv pass
            target = sv_ref()
            if target is not None:
                target.tensor

        rng.set_stats(None, hook=stats_with_sv)
        return sv, rng

    def _host_reach_bound(self, S):
        """EK only (CalibrateAZFP overrides it: its rows carry an offset r0 > 0).
        An upper bound of every echo_range the EK coefficient rows can produce -- fl((S - 1) max sample_interval)
        * (max sound_speed / 2), the row formula (range.py:138) at the two maxima -- from HOST copies of the two
        parameters (host arrays, or the mirrors EchoData.to_device keeps; memoised there), so that sizing the range
        grid costs neither a device reduction nor a wait for the GPU.  None when a parameter lives in HBM only."""
        def hmax(v):
            d = v.data if isinstance(v, DataArray) else None
            if isinstance(d, DeviceArray):
                return d.host_nanmax()
            a = np.asarray(getattr(v, "values", v), dtype=np.float64)
            with np.errstate(invalid="ignore"):
                m = float(np.fmax.reduce(a, axis=None)) if a.size else float("nan")
            return m if m == m else None

        env, beam = getattr(self, "env_params", None), getattr(self, "beam", None)
        if not env or beam is None or "sound_speed" not in env or "sample_interval" not in beam:
            return None
        si, cw = hmax(beam["sample_interval"]), hmax(env["sound_speed"])
        if si is None or cw is None or not (np.isfinite(si) and np.isfinite(cw) and si > 0 and cw > 0):
            return None
        # (float32 outputs: the range statistics the kernels leave are rounded to float32, up to 6e-8 above the float64
        #  value the bound is made from -- the margin keeps nanmax(echo_range) inside the conservative grid)
        margin = 1e-12 if str(getattr(self, "dtype", "float64")).endswith("64") else 1e-6
        return float((S - 1) * si) * (cw / 2) * (1 + margin)

    def _lazy_power_range(self, raw, coef, flags, stats=None):
        """echo_range of power samples as a LazyDeviceArray: coefficient rows + the raw samples' NaN pattern; written by
        epa_range_power on first read."""
        mask_flag = flags & _lib.FLAG_MASK_RANGE
        version, dtype = raw._version, self.dtype

        def make():
            if raw._version != version:
                raise RuntimeError("echo_range was left lazy by compute_Sv/compute_TS and backscatter_r has been "
                                   "modified in place since: its NaN mask can no longer be reproduced")
            return ops.range_power(raw, coef, flags=mask_flag, dtype=dtype)

        rng = LazyDeviceArray(tuple(raw.shape), dtype, raw.device, make, stats=stats, rows=coef,
                              nan_where=raw if mask_flag else None)
        rng.reach_bound = self._host_reach_bound(raw.shape[2])
        return rng
