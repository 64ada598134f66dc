"""Minimal labelled-array containers standing in for ``xarray.Dataset`` / ``DataArray``.

xarray is not installed in this image (SURVEY.md probe table), so the drop-in functions accept and
return these light containers; when real xarray IS importable they also accept ``xarray`` objects
(converted with :func:`from_xarray`) and can hand results back with :meth:`Dataset.to_xarray`.

Only what the hot path needs is implemented: named dimensions, coordinates, attributes, variable
access by name.  Array payloads may live on the host (numpy) or in HBM (:class:`DeviceArray`,
a lazily-copied torch CUDA tensor -- the analogue of the reference's dask-backed laziness:
``compute_Sv`` output stays on the device and feeds ``remove_background_noise`` /
``compute_MVBS`` without a PCIe round trip; ``.values`` materialises it on the host).
"""
from collections import OrderedDict

import numpy as np

try:  # pragma: no cover - not installed in the build image
    import xarray as _xr
except Exception:  # noqa: BLE001
    _xr = None

__all__ = ["DeviceArray", "LazyDeviceArray", "DataArray", "Dataset", "DeferredDataset", "from_xarray", "is_device", "is_xarray", "to_xarray", "xarray_io"]


class DeviceArray:
    """An array resident in GPU memory (wraps a torch CUDA tensor).  C-contiguous except for lazily transposed
    views (the impulse-noise mask); whoever hands the buffer to a kernel asks for ``.contiguous()`` first."""

    __slots__ = ("tensor", "_stats", "_host", "_hmax")

    def __init__(self, tensor, stats=None, host=None):
        self.tensor = tensor
        self._hmax = None
        # optional host copy the tensor was uploaded from (small parameter arrays, EchoData.to_device): host-side
        # decisions read it instead of copying the array back; void once the tensor has been modified in place
        self._host = (host, tensor._version) if host is not None else None
        # optional (f64 device tensor {nanmin, nanmax, NaN count}, tensor._version when they were taken): a
        # by-product of the kernel that wrote the array; void once the tensor has been modified in place
        self._stats = (stats, tensor._version) if stats is not None else None

    def cached_stats(self):
        """(nanmin, nanmax, nan_count) if the kernel that wrote this array left them and nothing touched it since."""
        if self._stats is None or self._stats[1] != self.tensor._version:
            return None
        lo, hi, nn = self._stats[0].cpu().tolist()
        return lo, hi, int(nn)

    def host_nanmax(self):
        """nanmax of the array taken from its host copy -- None without one (or when nothing is finite).  Memoised while
        the tensor is untouched: a host-side bound (how far can any echo_range reach?) then costs neither a device
        reduction nor a wait for the GPU."""
        if self._host is None or self._host[1] != self.tensor._version:
            return None
        if self._hmax is None or self._hmax[1] != self.tensor._version:
            a = self._host[0]
            with np.errstate(invalid="ignore"):
                m = float(np.fmax.reduce(a, axis=None)) if a.size else float("nan")
            self._hmax = (m, self.tensor._version)
        m = self._hmax[0]
        return m if m == m else None

    @property
    def shape(self):
        return tuple(self.tensor.shape)

    @property
    def ndim(self):
        return self.tensor.dim()

    @property
    def dtype(self):
        return np.dtype(str(self.tensor.dtype).replace("torch.", ""))

    @property
    def nbytes(self):
        return self.tensor.numel() * self.tensor.element_size()

    def __array__(self, dtype=None, copy=None):
        if self._host is not None and self._host[1] == self.tensor._version:
            a = self._host[0]
        else:
            a = self.tensor.detach().cpu().numpy()
        return a.astype(dtype) if dtype is not None else a

    def __repr__(self):
        return f"DeviceArray(shape={self.shape}, dtype={self.dtype}, device={self.tensor.device})"


class LazyDeviceArray(DeviceArray):
    """A device array that is written on first use (the analogue of a dask-backed variable of the reference):
    ``compute_Sv`` on power samples leaves ``echo_range`` in this form -- shape, dtype and the {nanmin, nanmax, NaN
    count} by-product are known, the 8 B/sample array itself is produced by ``make()`` when somebody reads ``.tensor``
    / ``.values``.  ``rows`` = the per-(channel, ping) coefficient rows the array is an affine function of
    (``range = fl(fl(s * ra) * rb) + r0``): kernels that accept those rows in place of the array (``compute_MVBS``)
    never need it.  The producer, the rows and the NaN source keep the raw samples and the coefficient rows alive for
    as long as this object lives (the echodata normally does anyway): replace the variable by
    ``DeviceArray(lazy.tensor)`` to cut that tie."""

    __slots__ = ("_make", "_shape", "_tdtype", "_device", "_tensor", "_rows", "_mask", "_made_version", "source",
                 "_stats_hook", "reach_bound", "_affine", "__weakref__")

    def __init__(self, shape, dtype, device, make, stats=None, rows=None, nan_where=None, source=None):
        self._make, self._shape, self._tdtype, self._device = make, tuple(int(n) for n in shape), dtype, device
        self._tensor = None
        self._hmax = None
        self._made_version = None  # tensor._version right after make(): any later in-place write voids rows and stats
        self._host = None
        self._stats = (stats, 0) if stats is not None else None
        self._rows = rows
        # what a deferred Sv is made from (calibrate_base.PowerSource): a consumer that can produce the array as a
        # by-product of its own pass over the same inputs (compute_MVBS: the fused kernel) does so and calls fulfil()
        self.source = source
        self._stats_hook = None  # called once when the statistics are asked for and nobody has left them yet
        # an upper bound of every value of the array known on the HOST (an echo_range: how far can any row reach?), or
        # None: a consumer sizes its launch from it and reads the exact statistics later (stats_async)
        self.reach_bound = None
        # (base lazy array, scale (C, P) f64 device tensor, offset likewise): this array is offset + scale * base, row by
        # row (consolidate.add_depth on a lazy echo_range) -- a kernel that evaluates the base from its coefficient rows
        # evaluates this one next to it (compute_MVBS(range_var="depth")); see affine_of()
        self._affine = None
        # the array is NaN exactly where this device tensor of the same shape is (the raw power samples): kernels that
        # need the NaN pattern as well as the values read it next to the rows
        self._mask = (nan_where, nan_where._version) if nan_where is not None else None

    def nan_source(self):
        """The tensor whose NaNs are the array's NaNs, if there is one and nothing has written to it since."""
        if self._mask is None or self._mask[0]._version != self._mask[1]:
            return None
        return self._mask[0]

    @property
    def materialized(self):
        return self._tensor is not None

    def fulfil(self, tensor):
        """Install the array produced elsewhere (same shape and dtype as make() would return)."""
        if self._tensor is not None:
            raise RuntimeError("the array has been written already")
        if tuple(tensor.shape) != self._shape or tensor.dtype != self._tdtype:
            raise ValueError(f"expected {self._shape} {self._tdtype}, got {tuple(tensor.shape)} {tensor.dtype}")
        self._tensor, self._make, self.source = tensor, None, None
        self._made_version = tensor._version
        if self._stats is not None:
            self._stats = (self._stats[0], tensor._version)

    def set_stats(self, stats, hook=None):
        """Leave the {nanmin, nanmax, NaN count} device tensor with the array (valid while nothing writes to it), or
        a ``hook`` that will (it must end up calling set_stats(stats))."""
        if stats is not None:
            self._stats = (stats, self._tensor._version if self._tensor is not None else 0)
            self._stats_hook = None
        else:
            self._stats_hook = hook

    @property
    def tensor(self):
        if self._tensor is None:
            t = self._make()
            if self._tensor is None:  # (make() may have gone through fulfil())
                self.fulfil(t)
        return self._tensor

    def coef_rows(self):
        """The coefficient rows, while the array still is the function of them it was created as."""
        if self._tensor is not None and self._made_version != self._tensor._version:
            return None
        return self._rows

    def set_affine(self, base, scale, offset):
        self._affine = (base, scale, offset, scale._version, offset._version)

    def affine_of(self):
        """(base, scale, offset) while this array still is ``offset + scale * base`` row by row and the base still is the
        function of its coefficient rows it was created as; None otherwise."""
        if self._affine is None or (self._tensor is not None and self._made_version != self._tensor._version):
            return None
        base, scale, offset, vs, vo = self._affine
        if scale._version != vs or offset._version != vo or base.coef_rows() is None:
            return None
        return base, scale, offset

    def stats_async(self):
        """The {nanmin, nanmax, NaN count} as an object whose ``.tolist()`` does not wait for kernels launched after
        this call (``ops.HostFuture``: the three numbers start for the host now, on a side stream), or None when
        nobody has left them (or something wrote to the array since)."""
        if self._stats is None or (self._tensor is not None and self._stats[1] != self._tensor._version):
            return None
        st = self._stats[0]
        if not hasattr(st, "tolist") or hasattr(st, "is_cuda"):  # a device tensor: send it on its way
            from . import ops

            st = ops.fetch_async(st)
            self._stats = (st, self._stats[1])
        return st

    def cached_stats(self):
        if self._stats is None and self._stats_hook is not None:
            hook, self._stats_hook = self._stats_hook, None
            hook()
        if self._stats is None or (self._tensor is not None and self._stats[1] != self._tensor._version):
            return None
        lo, hi, nn = self._stats[0].cpu().tolist()
        return lo, hi, int(nn)

    @property
    def shape(self):
        return self._shape

    @property
    def ndim(self):
        return len(self._shape)

    @property
    def dtype(self):
        return np.dtype(str(self._tdtype).replace("torch.", ""))

    @property
    def nbytes(self):
        return int(np.prod(self._shape)) * self.dtype.itemsize

    def __repr__(self):
        state = "materialized" if self.materialized else "lazy"
        return f"LazyDeviceArray(shape={self.shape}, dtype={self.dtype}, device={self._device}, {state})"


def is_device(a):
    return isinstance(a, DeviceArray)


def host_readable(a):
    """True when np.asarray(a) costs no device-to-host copy: a host array, or a DeviceArray that still holds the host
    copy it was uploaded from."""
    if not isinstance(a, DeviceArray):
        return True
    return a._host is not None and a._host[1] == a.tensor._version


class _Once:
    """A thunk evaluated once (shared by the copies of the attrs that carry it)."""

    __slots__ = ("fn", "value", "done")

    def __init__(self, fn):
        self.fn, self.value, self.done = fn, None, False

    def __call__(self):
        if not self.done:
            self.value, self.done, self.fn = self.fn(), True, None
        return self.value


_PENDING = type("Pending", (), {"__repr__": lambda self: "<value on its way from the GPU>"})()


class LazyAttrs(dict):
    """Attributes some of whose VALUES are numbers a kernel is still producing (``actual_range`` of
    ``remove_background_noise``'s outputs: their minimum / maximum): ``set_lazy(key, thunk)`` reserves the key in
    place, the thunk runs the first time anybody READS the value -- ``attrs[key]``, ``.get``, ``.items()``,
    ``.values()``, ``dict(attrs)``, ``{**attrs}``, ``==``, ``repr``, ``json.dumps`` -- and the result then sits in
    the dict like any other value.  Keys, ``len`` and ``in`` never wait.  ``copy()`` (what DataArray construction
    uses) keeps the thunks, shared: the value is computed once.  Everything else is a plain dict."""

    __slots__ = ("_pending",)

    def __init__(self, *a, **kw):
        self._pending = {}
        super().__init__()
        self.update(*a, **kw)

    def set_lazy(self, key, thunk):
        dict.__setitem__(self, key, _PENDING)
        self._pending[key] = thunk if isinstance(thunk, _Once) else _Once(thunk)

    def has_pending(self, key=None):
        return bool(self._pending) if key is None else key in self._pending

    def _settle(self, key=None):
        for k in ([key] if key is not None else list(self._pending)):
            th = self._pending.pop(k, None)
            if th is not None:
                dict.__setitem__(self, k, th())

    # -- reads of values settle them
    def __getitem__(self, key):
        self._settle(key)
        return dict.__getitem__(self, key)

    def get(self, key, default=None):
        self._settle(key)
        return dict.get(self, key, default)

    def pop(self, key, *default):
        self._settle(key)
        return dict.pop(self, key, *default)

    def popitem(self):
        self._settle()
        return dict.popitem(self)

    def setdefault(self, key, default=None):
        self._settle(key)
        return dict.setdefault(self, key, default)

    def items(self):
        self._settle()
        return dict.items(self)

    def values(self):
        self._settle()
        return dict.values(self)

    def __iter__(self):  # (overridden on purpose: dict(attrs) / {**attrs} then go through keys() + __getitem__)
        return dict.__iter__(self)

    def __eq__(self, other):
        self._settle()
        if isinstance(other, LazyAttrs):
            other._settle()
        return dict.__eq__(self, other)

    def __ne__(self, other):
        return not self.__eq__(other)

    __hash__ = None

    def __repr__(self):
        self._settle()
        return dict.__repr__(self)

    def __reduce__(self):
        self._settle()
        return (dict, (dict(dict.items(self)),))

    # -- writes drop what they replace
    def __setitem__(self, key, value):
        self._pending.pop(key, None)
        dict.__setitem__(self, key, value)

    def __delitem__(self, key):
        self._pending.pop(key, None)
        dict.__delitem__(self, key)

    def update(self, *a, **kw):
        for k, v in dict(*a, **kw).items():
            self[k] = v

    def clear(self):
        self._pending.clear()
        dict.clear(self)

    def copy(self):
        out = LazyAttrs()
        for k in dict.__iter__(self):
            if k in self._pending:
                out.set_lazy(k, self._pending[k])
            else:
                dict.__setitem__(out, k, dict.__getitem__(self, k))
        return out

    __copy__ = copy

    def __deepcopy__(self, memo):
        import copy as _copy

        self._settle()
        return _copy.deepcopy(dict(dict.items(self)), memo)


def _copy_attrs(attrs):
    """A new attrs dict; numbers still on their way from the GPU stay on their way."""
    return attrs.copy() if isinstance(attrs, LazyAttrs) else dict(attrs or {})


class DataArray:
    def __init__(self, data, dims=None, coords=None, attrs=None, name=None):
        if isinstance(data, DataArray):
            dims = dims if dims is not None else data.dims
            coords = coords if coords is not None else data.coords
            attrs = attrs if attrs is not None else data.attrs
            name = name if name is not None else data.name
            data = data.data
        if not isinstance(data, DeviceArray):
            data = np.asarray(data)
        self.data = data
        if dims is None:
            dims = () if data.ndim == 0 else tuple(f"dim_{i}" for i in range(data.ndim))
        if isinstance(dims, str):
            dims = (dims,)
        self.dims = tuple(dims)
        if len(self.dims) != data.ndim:
            raise ValueError(f"dims {self.dims} do not match data of shape {data.shape}")
        self.coords = OrderedDict()
        for k, v in (coords or {}).items():
            self.coords[k] = v if isinstance(v, np.ndarray) else np.asarray(getattr(v, "values", v))
        self.attrs = _copy_attrs(attrs)
        self.name = name

    # -- array protocol ---------------------------------------------------------------------
    @property
    def values(self):
        return np.asarray(self.data)

    @property
    def shape(self):
        return tuple(self.data.shape)

    @property
    def ndim(self):
        return self.data.ndim

    @property
    def dtype(self):
        return self.data.dtype

    @property
    def size(self):
        return int(np.prod(self.shape)) if self.shape else 1

    @property
    def sizes(self):
        return OrderedDict(zip(self.dims, self.shape))

    @property
    def nbytes(self):
        return self.data.nbytes

    def __array__(self, dtype=None, copy=None):
        a = self.values
        return a.astype(dtype) if dtype is not None else a

    def __getitem__(self, key):
        if isinstance(key, str):
            return DataArray(self.coords[key], dims=(key,) if self.coords[key].ndim == 1 else None)
        return self.values[key]

    def __len__(self):
        return self.shape[0]

    def copy(self):
        d = self.data if isinstance(self.data, DeviceArray) else self.data.copy()
        return DataArray(d, self.dims, dict(self.coords), self.attrs, self.name)

    def assign_attrs(self, attrs=None, **kw):
        out = DataArray(self.data, self.dims, dict(self.coords), self.attrs, self.name)
        out.attrs.update(attrs or {})
        out.attrs.update(kw)
        return out

    def isel(self, **indexers):
        a = self.values
        dims = list(self.dims)
        coords = dict(self.coords)
        for d, idx in indexers.items():
            ax = dims.index(d)
            a = np.take(a, idx, axis=ax) if not isinstance(idx, slice) else a[(slice(None),) * ax + (idx,)]
            if d in coords:
                coords[d] = coords[d][idx]
            if np.isscalar(idx) or isinstance(idx, (int, np.integer)):
                dims.pop(ax)
        coords = {k: v for k, v in coords.items() if k in dims}
        return DataArray(a, dims, coords, self.attrs, self.name)

    def isnull(self):
        return DataArray(np.isnan(self.values), self.dims, dict(self.coords))

    def min(self):
        return np.nanmin(self.values)

    def max(self):
        return np.nanmax(self.values)

    def __repr__(self):
        return f"<DataArray {self.name!r} {dict(self.sizes)} {self.dtype} on {'device' if is_device(self.data) else 'host'}>"


class Dataset:
    def __init__(self, data_vars=None, coords=None, attrs=None):
        self.coords = OrderedDict()
        self.data_vars = OrderedDict()
        self.attrs = dict(attrs or {})
        for k, v in (coords or {}).items():
            self._set_coord(k, v)
        for k, v in (data_vars or {}).items():
            self[k] = v

    def _set_coord(self, name, v):
        if isinstance(v, DataArray):
            arr, attrs = v.values, v.attrs
        elif isinstance(v, tuple):
            arr, attrs = np.asarray(v[1]), (v[2] if len(v) > 2 else {})
        else:
            arr, attrs = np.asarray(getattr(v, "values", v)), {}
        self.coords[name] = DataArray(arr, (name,) if arr.ndim == 1 else None, attrs=attrs, name=name)

    # -- mapping protocol -------------------------------------------------------------------
    def __setitem__(self, name, v):
        if isinstance(v, tuple):
            dims, data = v[0], v[1]
            attrs = v[2] if len(v) > 2 else {}
            da = DataArray(data, dims, attrs=attrs, name=name)
        elif isinstance(v, DataArray):
            da = DataArray(v.data, v.dims, v.coords, v.attrs, name)
        else:
            arr = np.asarray(v)
            da = DataArray(arr, () if arr.ndim == 0 else None, name=name)
        for d, n in zip(da.dims, da.shape):
            if d in self.coords and self.coords[d].shape[0] != n:
                raise ValueError(f"variable {name!r}: dimension {d!r} has length {n}, dataset has "
                                 f"{self.coords[d].shape[0]}")
        for k, c in da.coords.items():
            if k not in self.coords and np.ndim(c) == 1:
                self._set_coord(k, c)
        if name in self.coords and da.dims == (name,):
            self._set_coord(name, da)
            return
        self.data_vars[name] = da

    def __getitem__(self, name):
        if name in self.data_vars:
            da = self.data_vars[name]
            cs = {d: self.coords[d].values for d in da.dims if d in self.coords}
            return DataArray(da.data, da.dims, cs, da.attrs, name)
        if name in self.coords:
            return self.coords[name]
        raise KeyError(name)

    def __contains__(self, name):
        return name in self.data_vars or name in self.coords

    def __iter__(self):
        return iter(self.data_vars)

    def get(self, name, default=None):
        return self[name] if name in self else default

    @property
    def variables(self):
        out = OrderedDict(self.coords)
        out.update(self.data_vars)
        return out

    @property
    def sizes(self):
        out = OrderedDict()
        for da in self.data_vars.values():
            for d, n in zip(da.dims, da.shape):
                out.setdefault(d, n)
        for k, c in self.coords.items():
            if c.ndim == 1:
                out.setdefault(k, c.shape[0])
        return out

    dims = sizes

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError(name)
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name) from None

    def copy(self):
        out = Dataset(attrs=dict(self.attrs))
        for k, c in self.coords.items():
            out.coords[k] = c
        for k, v in self.data_vars.items():
            out.data_vars[k] = v
        return out

    def assign_attrs(self, attrs=None, **kw):
        out = self.copy()
        out.attrs.update(attrs or {})
        out.attrs.update(kw)
        return out

    def drop_vars(self, names):
        names = [names] if isinstance(names, str) else list(names)
        out = self.copy()
        for n in names:
            out.data_vars.pop(n, None)
            out.coords.pop(n, None)
        return out

    def isel(self, **indexers):
        out = Dataset(attrs=dict(self.attrs))
        for k, c in self.coords.items():
            out.coords[k] = c.isel(**{d: i for d, i in indexers.items() if d in c.dims}) if any(
                d in c.dims for d in indexers) else c
        for k, v in self.data_vars.items():
            sel = {d: i for d, i in indexers.items() if d in v.dims}
            out.data_vars[k] = v.isel(**sel) if sel else v
        return out

    def to_host(self):
        """Materialise every device-resident variable on the host (PCIe copies)."""
        out = self.copy()
        for k, v in list(out.data_vars.items()):
            if is_device(v.data):
                out.data_vars[k] = DataArray(v.values, v.dims, v.coords, v.attrs, k)
        return out

    def to_xarray(self):
        if _xr is None:
            raise ImportError("xarray is not installed")
        return _xr.Dataset(
            {k: (v.dims, v.values, v.attrs) for k, v in self.data_vars.items()},
            coords={k: (c.dims, c.values, c.attrs) for k, c in self.coords.items()}, attrs=self.attrs)

    def __repr__(self):
        lines = [f"<Dataset {dict(self.sizes)}>"]
        lines += [f"  coord {k}: {c.shape} {c.dtype}" for k, c in self.coords.items()]
        lines += [f"  {v!r}" for v in self.data_vars.values()]
        return "\n".join(lines)


class DeferredDataset(Dataset):
    """A Dataset whose assembly waits for its first use (the analogue of a dask-backed result of the reference).

    ``compute_MVBS`` right after ``compute_Sv`` launches ONE kernel that writes the Sv array and the bins; the shape of
    the result -- ``np.arange(0, nanmax(echo_range) + bin, bin)`` (commongrid/api.py:108-115) -- depends on a number
    that kernel produces.  Waiting for it inside the call would stall the host for the ~10 ms the kernel runs; instead
    the call returns this object at once, and ``build()`` (read the three statistics back, trim the grid, attach
    coordinates / attributes) runs when anybody touches it: ``.coords``, ``.data_vars``, ``.attrs``, ``ds["Sv"]``,
    ``.sizes``, ``repr`` ...  A pipeline that calibrates file k + 1 before it looks at the MVBS of file k keeps the GPU
    busy back to back.  Errors and warnings the assembly raises (an empty range grid, NaN coordinates) surface at that
    first use.  ``EPA_DEFER_MVBS=0`` makes the calls assemble before they return."""

    def __init__(self, build):  # (Dataset.__init__ is not run: the three containers are the built dataset's)
        self.__dict__["_build"] = build
        self.__dict__["_ds"] = None
        self.__dict__["_error"] = None

    def _resolve(self):
        d = self.__dict__
        if d["_ds"] is None:
            build, d["_build"] = d["_build"], None
            if build is None:  # the assembly failed at an earlier access: the same error again, not a generic one
                raise d["_error"]
            try:
                d["_ds"] = build()
            except Exception as e:  # kept: every later access raises the same error
                d["_error"] = e
                raise
            except BaseException:  # an interrupt is not the assembly's verdict: the next access tries again
                d["_build"] = build
                raise
        return d["_ds"]

    @property
    def resolved(self):
        return self.__dict__["_ds"] is not None

    coords = property(lambda self: self._resolve().coords, lambda self, v: setattr(self._resolve(), "coords", v))
    data_vars = property(lambda self: self._resolve().data_vars, lambda self, v: setattr(self._resolve(), "data_vars", v))
    attrs = property(lambda self: self._resolve().attrs, lambda self, v: setattr(self._resolve(), "attrs", v))

    def __repr__(self):
        return Dataset.__repr__(self._resolve())

    def __copy__(self):  # (a shallow copy of the proxy would assemble twice)
        return self._resolve().copy()

    def __deepcopy__(self, memo):
        import copy

        return copy.deepcopy(self._resolve(), memo)


def defer_mvbs_enabled():
    import os

    return os.environ.get("EPA_DEFER_MVBS", "1") != "0"


def from_xarray(obj):
    """xarray.Dataset / DataArray -> lite container (no-op for lite containers)."""
    if isinstance(obj, (Dataset, DataArray)) or _xr is None:
        return obj
    if isinstance(obj, _xr.DataArray):
        return DataArray(obj.values, obj.dims, {k: v.values for k, v in obj.coords.items() if v.ndim == 1},
                         dict(obj.attrs), obj.name)
    if isinstance(obj, _xr.Dataset):
        ds = Dataset(attrs=dict(obj.attrs))
        for k, c in obj.coords.items():
            if c.ndim == 1 and c.dims == (k,):
                ds._set_coord(k, (c.dims, c.values, dict(c.attrs)))
        for k, v in obj.data_vars.items():
            ds[k] = (v.dims, v.values, dict(v.attrs))
        return ds
    return obj


# ---- the xarray boundary: xarray in -> xarray out ------------------------------------------------------------------
def is_xarray(obj):
    return _xr is not None and isinstance(obj, (_xr.Dataset, _xr.DataArray))


def to_xarray(obj):
    """lite Dataset / DataArray (or a tuple / list of them) -> xarray objects; anything else passes through."""
    if isinstance(obj, Dataset):
        return obj.to_xarray()
    if isinstance(obj, DataArray):
        if _xr is None:
            raise ImportError("xarray is not installed")
        return _xr.DataArray(obj.values, dims=obj.dims, coords={k: v for k, v in obj.coords.items() if np.ndim(v) == 1},
                             attrs=dict(obj.attrs), name=obj.name)
    if isinstance(obj, (tuple, list)):
        return type(obj)(to_xarray(o) for o in obj)
    return obj


def xarray_io(in_place=()):
    """Decorator of the drop-in API functions (first argument = the dataset / EchoData the reference takes).

    Lite containers in -> lite containers out (results stay in HBM).  A real ``xarray`` Dataset / DataArray in, or an
    EchoData whose groups are xarray Datasets (echopype's own): the function runs on the converted copy and its
    results are handed back as ``xarray`` objects (``.values`` of device arrays are copied to the host once).
    ``in_place``: names of variables the reference ADDS TO THE CALLER'S dataset (clean/api.py:490-502 assigns
    ``ds_Sv["Sv_noise"]`` / ``ds_Sv["Sv_corrected"]`` before it returns): they are written into the caller's xarray
    dataset too, and that dataset (with the updated attributes) is returned."""
    import functools
    import inspect

    def deco(fn):
        params = list(inspect.signature(fn).parameters)
        first_name = params[0]
        # other arguments that may be foreign containers too (add_depth(ds, echodata=ed), apply_mask(source_ds, mask))
        ed_kw = "echodata" if "echodata" in params[1:] else None

        @functools.wraps(fn)
        def wrapper(*args, **kwargs):
            from .echodata import EchoData, as_lite_echodata

            if args:
                first, args = args[0], args[1:]
            elif first_name in kwargs:  # the reference's functions take their dataset by keyword as well
                first = kwargs.pop(first_name)
            else:
                raise TypeError(f"{fn.__name__}() missing 1 required positional argument: '{first_name}'")
            if ed_kw and kwargs.get(ed_kw) is not None and not isinstance(kwargs[ed_kw], EchoData) \
                    and hasattr(kwargs[ed_kw], "sonar_model"):
                kwargs[ed_kw] = as_lite_echodata(kwargs[ed_kw])
            foreign_ed = not isinstance(first, EchoData) and hasattr(first, "sonar_model")
            if not (is_xarray(first) or foreign_ed):
                return fn(first, *args, **kwargs)
            lite = as_lite_echodata(first) if foreign_ed else from_xarray(first)
            out = fn(lite, *args, **kwargs)
            if in_place and isinstance(out, Dataset) and is_xarray(first):
                for name in in_place:
                    if name in out.data_vars:
                        v = out.data_vars[name]
                        first[name] = (v.dims, v.values, dict(v.attrs))
                return first.assign_attrs(dict(out.attrs))
            return to_xarray(out)

        return wrapper

    return deco
