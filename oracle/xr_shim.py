"""A minimal named-dimension array (the subset of xarray semantics the reference's calibrator methods
use) -- TEST INFRASTRUCTURE for oracle/gen_chain_goldens.py, authoring container only.

xarray is not installable here, so the reference's array-level methods (calibrate/range.py,
CalibrateEK._cal_power_samples, CalibrateAZFP._cal_power_samples) cannot run as they are.  Their
bodies are plain arithmetic on labelled arrays; this shim gives them exactly that: arrays that
broadcast BY DIMENSION NAME (result dims = dims of the left operand, then the new ones of the right,
as xarray orders them), NumPy ufuncs applied element-wise, ``where`` / ``isnull`` / ``isel`` /
``transpose`` / boolean selection along one dimension.  Label alignment is emulated only where the
reference relies on it (inner join of a channel subset, range.py:199); otherwise operands that share
a dimension must share its length (asserted).
It is deliberately small and strict: anything the reference calls that is not implemented raises
AttributeError, so a silent semantic difference cannot slip in.
"""
import numpy as np

__all__ = ["DataArray", "Dataset", "where", "merge", "apply_ufunc", "concat", "full_like", "zeros_like"]


def _as_da(x):
    return x if isinstance(x, DataArray) else None


class _Coords(dict):
    pass


class DataArray:
    __array_priority__ = 1000

    def __init__(self, data=None, coords=None, dims=None, name=None, attrs=None):
        if isinstance(data, DataArray):
            dims = data.dims if dims is None else dims
            coords = data.coords if coords is None else coords
            name = data.name if name is None else name
            attrs = data.attrs if attrs is None else attrs
            data = data.data
        self.data = np.asarray(data)
        if dims is None:
            if isinstance(coords, (list, tuple)) and coords and isinstance(coords[0], tuple):
                dims = [c[0] for c in coords]
            elif isinstance(coords, (list, tuple)) and coords and isinstance(coords[0], DataArray):
                dims = [c.dims[0] for c in coords]
            elif isinstance(coords, dict) and self.data.ndim == len(coords):
                dims = list(coords)
            else:
                dims = [f"dim_{i}" for i in range(self.data.ndim)]
        if isinstance(dims, str):
            dims = [dims]
        self.dims = tuple(dims)
        assert len(self.dims) == self.data.ndim, (self.dims, self.data.shape)
        self.coords = _Coords()
        if isinstance(coords, (list, tuple)) and coords and isinstance(coords[0], DataArray):
            coords = {d: c for d, c in zip(dims, coords)}
        if isinstance(coords, dict):
            for k, v in coords.items():
                v = v.data if isinstance(v, DataArray) else (v[1] if isinstance(v, tuple) else v)
                self.coords[k] = np.asarray(v)
        self.name = name
        self.attrs = dict(attrs or {})

    # ---- basic protocol
    @property
    def values(self):
        return self.data

    @values.setter
    def values(self, v):  # replaces the variable's data, no write-through to the array it was sliced from
        v = np.asarray(v)
        assert v.shape == self.data.shape
        self.data = v

    def __bool__(self):
        assert self.data.size == 1
        return bool(self.data.reshape(()))

    def __float__(self):  # xarray defines the scalar conversions for one-element arrays
        return float(self.data.reshape(()))

    def __int__(self):
        return int(self.data.reshape(()))

    def chunk(self, *a, **k):
        return self

    @property
    def shape(self):
        return self.data.shape

    @property
    def ndim(self):
        return self.data.ndim

    @property
    def dtype(self):
        return self.data.dtype

    @property
    def sizes(self):
        return dict(zip(self.dims, self.data.shape))

    def __len__(self):
        return self.data.shape[0]

    def __array__(self, dtype=None, copy=None):
        return self.data.astype(dtype) if dtype is not None else self.data

    def __repr__(self):
        return f"<shim.DataArray {self.name!r} {self.sizes}>"

    def compute(self):
        return self

    def copy(self):
        return DataArray(self.data.copy(), dict(self.coords), self.dims, self.name, dict(self.attrs))

    def astype(self, dt):
        return self._like(self.data.astype(dt))

    def _like(self, data, dims=None):
        dims = self.dims if dims is None else tuple(dims)
        return DataArray(data, {k: v for k, v in self.coords.items() if k in dims and np.ndim(v) == 1}, dims, self.name,
                         self.attrs)

    # ---- broadcasting by dimension name
    def _expand(self, dims_out):
        perm = [self.dims.index(d) for d in dims_out if d in self.dims]
        a = np.transpose(self.data, perm)
        idx = tuple(slice(None) if d in self.dims else None for d in dims_out)
        return a[idx]

    @staticmethod
    def _union(a, b):
        dims = list(a.dims) + [d for d in b.dims if d not in a.dims]
        sa, sb = a.sizes, b.sizes
        for d in dims:
            if d in sa and d in sb:
                assert sa[d] == sb[d], f"dimension {d!r}: {sa[d]} vs {sb[d]} (no label alignment in the shim)"
        return dims

    @staticmethod
    def _align(a, b):
        """xarray's default inner join, only for a shared dimension whose lengths differ and whose
        labels both operands carry (range.py:199: all channels vs the GPT channels)."""
        for d in a.dims:
            if d in b.dims and a.sizes[d] != b.sizes[d]:
                la, lb = list(a.coords[d]), list(b.coords[d])
                common = [x for x in la if x in lb]
                a = a.isel(**{d: np.array([la.index(x) for x in common])})
                b = b.isel(**{d: np.array([lb.index(x) for x in common])})
            elif d in b.dims and d in a.coords and d in b.coords:
                # equal lengths, both labelled: xarray joins on the LABELS all the same (default join="inner":
                # PandasIndex.join -> pandas Index.intersection, which keeps the order of the FIRST operand).  Equal
                # labels combine by position; a permutation of the same unique labels is re-ordered through pandas
                # itself; anything else (labels dropped by the join) must fail loudly, not silently
                la, lb = np.asarray(a.coords[d]), np.asarray(b.coords[d])
                same = la.shape == lb.shape and bool(np.all((la == lb) | ((la != la) & (lb != lb)))) \
                    if la.dtype.kind not in "OUS" else list(la) == list(lb)
                if not same:
                    import pandas as pd

                    ia, ib = pd.Index(la), pd.Index(lb)
                    common = ia.intersection(ib)
                    assert ia.is_unique and ib.is_unique and len(common) == len(ia), \
                        f"operands carry different {d!r} labels: an inner join that drops labels is beyond the shim"
                    a = a.isel(**{d: ia.get_indexer(common)})
                    b = b.isel(**{d: ib.get_indexer(common)})
        return a, b

    def _binary(self, other, f, reflexive=False):
        o = _as_da(other)
        if o is None:
            if isinstance(other, np.ndarray) and other.ndim > 0 and other.shape != self.data.shape:
                # (an ndarray of exactly the array's shape combines positionally, as in xarray)
                raise TypeError("unlabelled ndarray operand of a different shape")
            r = f(other, self.data) if reflexive else f(self.data, other)
            return self._like(r)
        self, o = DataArray._align(self, o)
        dims = self._union(self, o) if not reflexive else self._union(o, self)
        x, y = self._expand(dims), o._expand(dims)
        r = f(y, x) if reflexive else f(x, y)
        coords = {**{k: v for k, v in o.coords.items() if np.ndim(v) == 1},
                  **{k: v for k, v in self.coords.items() if np.ndim(v) == 1}}
        return DataArray(r, {k: v for k, v in coords.items() if k in dims}, dims)

    def __array_ufunc__(self, ufunc, method, *inputs, **kw):
        if method != "__call__" or kw.get("out") is not None:
            return NotImplemented
        if len(inputs) == 1:
            return self._like(ufunc(self.data, **kw))
        if len(inputs) == 2:
            a, b = inputs
            if a is self:
                return self._binary(b, lambda x, y: ufunc(x, y, **kw))
            return self._binary(a, lambda x, y: ufunc(x, y, **kw), reflexive=True)
        return NotImplemented


def _op(name, f):
    def fwd(self, other):
        return self._binary(other, f)

    def rev(self, other):
        return self._binary(other, f, reflexive=True)

    setattr(DataArray, f"__{name}__", fwd)
    setattr(DataArray, f"__r{name}__", rev)


for _n, _f in (("add", np.add), ("sub", np.subtract), ("mul", np.multiply), ("truediv", np.true_divide),
               ("pow", np.power)):
    _op(_n, _f)
for _n, _f in (("gt", np.greater), ("lt", np.less), ("ge", np.greater_equal), ("le", np.less_equal),
               ("eq", np.equal), ("ne", np.not_equal), ("and", np.logical_and), ("or", np.logical_or)):
    setattr(DataArray, f"__{_n}__", (lambda f: lambda self, other: self._binary(other, f))(_f))
DataArray.__hash__ = None
DataArray.__neg__ = lambda self: self._like(-self.data)
DataArray.__invert__ = lambda self: self._like(~self.data)
DataArray.__contains__ = lambda self, v: bool(np.any(self.data == v))


def _methods():
    def isnull(self):
        return self._like(np.isnan(self.data))

    def where(self, cond, other=np.nan):
        c = cond if isinstance(cond, DataArray) else DataArray(cond, dims=self.dims)
        dims = DataArray._union(self, c)
        r = np.where(c._expand(dims), self._expand(dims), other)
        return self._like(r, dims)

    def transpose(self, *dims):
        dims = [d for d in dims if d in self.dims] + [d for d in self.dims if d not in dims]
        return self._like(np.transpose(self.data, [self.dims.index(d) for d in dims]), dims)

    def isel(self, **ix):
        a, dims = self.data, list(self.dims)
        coords = {k: v for k, v in self.coords.items() if np.ndim(v) == 1}
        for d, i in ix.items():
            ax = dims.index(d)
            a = a[(slice(None),) * ax + (i,)]
            if d in coords:
                coords[d] = coords[d][i]
            if not isinstance(i, slice) and np.ndim(i) == 0:  # integer index drops the dimension
                dims.pop(ax)
                coords.pop(d, None)
        return DataArray(a, coords, dims, self.name, self.attrs)

    def drop_vars(self, names, errors="raise"):
        names = [names] if isinstance(names, str) else names
        out = self.copy()
        for n in names:
            out.coords.pop(n, None)
        return out

    def squeeze(self):
        keep = [d for d, n in self.sizes.items() if n != 1]
        return self._like(self.data.reshape([self.sizes[d] for d in keep]), keep)

    def to_dataset(self):
        ds = Dataset()
        ds[self.name] = self
        return ds

    def getitem(self, key):
        if isinstance(key, str):  # a coordinate comes back as an array labelled by itself, as in xarray
            return DataArray(self.coords[key], {key: self.coords[key]}, [key], key)
        k = _as_da(key)
        if k is not None and k.data.dtype == bool and k.ndim == 1:  # boolean selection along its dimension
            ax = self.dims.index(k.dims[0])
            out = np.compress(k.data, self.data, axis=ax)
            coords = {c: (v[k.data] if c == k.dims[0] else v) for c, v in self.coords.items() if np.ndim(v) == 1}
            return DataArray(out, coords, self.dims, self.name, self.attrs)
        raise NotImplementedError(f"DataArray[{key!r}]")

    def setitem(self, key, value):
        if isinstance(key, dict):  # da[dict(dim=int, ...)] = scalar
            idx = tuple(key.get(d, slice(None)) for d in self.dims)
            self.data[idx] = value.data if isinstance(value, DataArray) else value
            return
        k = _as_da(key)
        if k is not None and k.data.dtype == bool and k.ndim == 1:
            ax = self.dims.index(k.dims[0])
            v = value.data if isinstance(value, DataArray) else np.asarray(value)
            idx = (slice(None),) * ax + (k.data,)
            self.data = self.data.copy()
            self.data[idx] = v
            return
        raise NotImplementedError(f"DataArray[{key!r}] = ...")

    def sel(self, drop=False, **ix):
        out = self
        for d, k in ix.items():
            if isinstance(k, slice):
                # label slice on a monotonic index: xarray hands it to pandas' Index.slice_indexer (both ends
                # inclusive); pandas itself is executed here
                import pandas as pd

                assert k.step is None
                idx = pd.Index(np.asarray(out.coords[d]))
                assert idx.is_monotonic_increasing, f"sel(slice) on a non-monotonic {d}"
                out = out.isel(**{d: idx.slice_indexer(k.start, k.stop)})
                continue
            k = _as_da(k)
            if k is not None and k.data.dtype != bool:
                # vectorised (pointwise) selection by LABEL: the indexer's own dimensions are shared with the array,
                # out[..., i, ...] = array[..., label -> position along d, ...] (xarray: the result drops d)
                assert all(dd in out.dims for dd in k.dims) and d not in k.dims
                labels = list(np.asarray(out.coords[d]))
                pos = np.vectorize(lambda v: labels.index(v))(k.data)
                for dd in k.dims:  # the shared dimensions must carry the same labels (xarray checks the indexer's)
                    if dd in k.coords and dd in out.coords:
                        assert list(np.asarray(k.coords[dd])) == list(np.asarray(out.coords[dd])), dd
                rest = [x for x in out.dims if x != d]
                src = np.transpose(out.data, [out.dims.index(d)] + [out.dims.index(x) for x in rest])
                kk = DataArray(pos, dims=k.dims)._expand(rest)
                kk = np.broadcast_to(kk, src.shape[1:])
                res = np.take_along_axis(src, kk[None], axis=0)[0]
                cs = {c: v for c, v in out.coords.items() if c != d and np.ndim(v) == 1}
                if not ix.get("drop", False):
                    pass
                out = DataArray(res, cs, rest, out.name, out.attrs)
                continue
            if k is None:
                raise NotImplementedError("sel: label slices, boolean and label DataArray indexers only")
            out = out.isel(**{d: np.flatnonzero(k.data)})
        return out

    class _Resample:
        """da.resample(time=freq).first().indexes[time]: the bin LABELS only.  xarray groups with a
        pandas Grouper(freq, closed / label / origin at pandas' defaults), so the labels are those of a pandas
        resample of the same index -- pandas is executed."""

        def __init__(self, da, dim, freq):
            import pandas as pd

            t = pd.DatetimeIndex(np.asarray(da.coords[dim]))
            self.dim = dim
            self.index = pd.Series(np.arange(len(t)), index=t).resample(freq).first().index

        def first(self):
            return self

        @property
        def indexes(self):
            return {self.dim: self.index}

    def resample(self, skipna=None, **kw):
        (dim, freq), = kw.items()
        return _Resample(self, dim, freq)

    def diff(self, dim, n=1, label="upper"):
        assert n == 1
        ax = self.dims.index(dim)
        out = np.diff(self.data, axis=ax)
        res = self._like(out)
        if dim in self.coords:
            c = np.asarray(self.coords[dim])
            res.coords[dim] = c[:-1] if label == "lower" else c[1:]
        return res

    class _Loc:
        def __init__(self, da):
            self.da = da

        def __setitem__(self, key, value):
            (d, k), = key.items()
            idx = np.flatnonzero(_as_da(k).data)
            v = value.transpose(*self.da.dims).data if isinstance(value, DataArray) else value
            ax = self.da.dims.index(d)
            self.da.data = self.da.data.copy()
            self.da.data[(slice(None),) * ax + (idx,)] = v

    def mean(self, dim=None, skipna=None):
        """xarray's default: NaN-skipping for float and complex data (duck_array_ops.mean: dtype.kind in "cfO"), plain
        otherwise; an all-NaN slice gives NaN."""
        import warnings

        skip = skipna if skipna is not None else self.data.dtype.kind in "fc"
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", RuntimeWarning)
            return self._reduce(np.nanmean if skip else np.mean, dim)

    def fillna(self, v):
        return self._like(np.where(np.isnan(self.data), v, self.data))

    def _reduce(self, f, dim):
        if dim is None:
            return DataArray(f(self.data, axis=None), dims=[])
        ax = self.dims.index(dim)
        return self._like(f(self.data, axis=ax), [d for d in self.dims if d != dim])

    def amin(self, dim=None, skipna=True):  # xarray skips NaN for float data by default
        import warnings

        with warnings.catch_warnings():
            warnings.simplefilter("ignore", RuntimeWarning)
            return self._reduce(np.nanmin if skipna else np.min, dim)

    def amax(self, dim=None, skipna=True):
        import warnings

        with warnings.catch_warnings():
            warnings.simplefilter("ignore", RuntimeWarning)
            return self._reduce(np.nanmax if skipna else np.max, dim)

    def assign_coords(self, **kw):
        out = self.copy()
        for k, v in kw.items():
            out.coords[k] = np.asarray(v.data if isinstance(v, DataArray) else v)
            assert k not in out.dims or len(out.coords[k]) == out.sizes[k]
        return out

    def reindex(self, indexers, method=None):
        """Forward-fill reindex along one sorted integer coordinate (clean/api.py:425-428)."""
        (dim, target), = indexers.items()
        assert method == "ffill"
        src = np.asarray(self.coords[dim])
        tgt = np.asarray(target.data if isinstance(target, DataArray) else target)
        pos = np.searchsorted(src, tgt, side="right") - 1
        ax = self.dims.index(dim)
        out = np.take(self.data, np.clip(pos, 0, None), axis=ax).astype(float)
        out[(slice(None),) * ax + (pos < 0,)] = np.nan
        res = self._like(out)
        res.coords[dim] = tgt
        return res

    def reindex_like(self, other):
        """Conform to ``other``'s labels along every shared dimension; labels this array lacks become NaN."""
        out = self
        for d in self.dims:
            src, tgt = np.asarray(out.coords[d]), np.asarray(other.coords[d])
            if len(src) == len(tgt) and np.array_equal(src, tgt):
                continue
            pos = {v: i for i, v in enumerate(src.tolist())}
            take = np.array([pos.get(v, -1) for v in tgt.tolist()])
            ax = out.dims.index(d)
            a = np.take(out.data, np.clip(take, 0, None), axis=ax).astype(float)
            a[(slice(None),) * ax + (take < 0,)] = np.nan
            out = out._like(a)
            out.coords[d] = tgt
        return out

    class _Coarsen:
        def __init__(self, da, windows, boundary):
            assert boundary == "pad"
            self.da, self.windows = da, windows

        def mean(self, skipna=True):
            """All window axes are reduced TOGETHER (xarray reshapes every coarsened dimension into
            (blocks, window) and reduces over the window axes at once): a NaN-skipping mean over the 2-D
            block, not a mean of per-row means."""
            import warnings

            a, shape, red = self._blocks()  # boundary="pad": NaN padding up to a whole number of windows
            with warnings.catch_warnings():
                warnings.simplefilter("ignore", RuntimeWarning)
                a = (np.nanmean if skipna else np.mean)(a.reshape(shape), axis=tuple(red))
            return self._with_coords(a)

        def min(self, skipna=True):
            import warnings

            a, shape, red = self._blocks()
            with warnings.catch_warnings():
                warnings.simplefilter("ignore", RuntimeWarning)
                a = (np.nanmin if skipna else np.min)(a.reshape(shape), axis=tuple(red))
            return self._with_coords(a)

        def _blocks(self):
            a, dims = self.da.data.astype(float), self.da.dims
            shape, red = [], []
            for ax, d in enumerate(dims):
                n = self.windows.get(d)
                if n is None:
                    shape.append(a.shape[ax])
                    continue
                padn = (-a.shape[ax]) % n
                if padn:
                    pad = [(0, 0)] * a.ndim
                    pad[ax] = (0, padn)
                    a = np.pad(a, pad, constant_values=np.nan)
                shape += [a.shape[ax] // n, n]
                red.append(len(shape) - 1)
            return a, shape, red

        def _with_coords(self, a):
            """Coordinates of a coarsened dimension: xarray's default ``coord_func="mean"`` -- the NaN / NaT
            skipping mean of the labels in each window (datetime64: float mean of the offsets from the
            earliest label, truncated to whole ns, duck_array_ops.mean)."""
            import warnings

            dims = self.da.dims
            out = DataArray(a, dims=dims, name=self.da.name, attrs=self.da.attrs)
            for d in dims:
                if d not in self.da.coords:
                    continue
                lab = np.asarray(self.da.coords[d])
                n = self.windows.get(d)
                if n is None:
                    out.coords[d] = lab
                    continue
                padn = (-len(lab)) % n
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore", RuntimeWarning)
                    if lab.dtype.kind == "M":
                        lab = lab.astype("datetime64[ns]")
                        off = np.nanmin(lab)
                        rel = np.where(np.isnat(lab), np.nan, (lab - off).astype("timedelta64[ns]").astype(float))
                        rel = np.pad(rel, (0, padn), constant_values=np.nan).reshape(-1, n)
                        out.coords[d] = np.nanmean(rel, axis=1).astype("timedelta64[ns]") + off
                    else:
                        rel = np.pad(lab.astype(float), (0, padn), constant_values=np.nan).reshape(-1, n)
                        out.coords[d] = np.nanmean(rel, axis=1)
            return out

    def coarsen(self, boundary="exact", **windows):
        return _Coarsen(self, windows, boundary)

    def pipe(self, f, *a, **k):
        return f(self, *a, **k)

    def any_(self):
        return bool(np.any(self.data))

    def rename(self, mapping):
        dims = tuple(mapping.get(d, d) for d in self.dims)
        coords = {mapping.get(k, k): v for k, v in self.coords.items()}
        return DataArray(self.data, coords, dims, mapping.get(self.name, self.name), self.attrs)

    def equals(self, other):
        """Same dims, same values (NaN == NaN), same coordinate labels."""
        if not isinstance(other, DataArray) or self.dims != other.dims or self.shape != other.shape:
            return False
        a, b = self.data, other.data
        same = (a == b) | ((a != a) & (b != b)) if a.dtype.kind == "f" else (a == b)
        if not bool(np.all(same)):
            return False
        return all(k in other.coords and np.array_equal(v, other.coords[k]) for k, v in self.coords.items()
                   if np.ndim(v) == 1)

    def assign_attrs(self, attrs=None, **kw):
        out = self.copy()
        out.attrs.update({**(attrs or {}), **kw})
        return out

    def iterate(self):
        assert self.ndim == 1
        for i in range(self.data.shape[0]):
            yield DataArray(self.data[i], dims=[], name=self.name)

    def idxmin(self, dim, skipna=True):
        """Label of the minimum along ``dim`` (NaN skipped; an all-NaN slice gives NaN, as xarray's idxmin)."""
        ax = self.dims.index(dim)
        a = self.data.astype(np.float64)
        allnan = np.all(np.isnan(a), axis=ax)
        pos = np.argmin(np.where(np.isnan(a), np.inf, a), axis=ax)  # first minimum, as nanargmin
        lab = np.asarray(self.coords[dim])[pos]
        lab = np.where(allnan, np.nan, lab.astype(np.float64)) if allnan.any() else lab
        return self._like(lab, [d for d in self.dims if d != dim])

    def sortby(self, key, ascending=True):
        k = key if isinstance(key, DataArray) else self[key]
        assert k.ndim == 1 and k.dims[0] in self.dims
        order = np.argsort(np.asarray(k.data), kind="stable")
        return self.isel(**{k.dims[0]: order if ascending else order[::-1]})

    def expand_dims(self, dim=None, **kw):
        """expand_dims(name=labels): a new leading dimension carrying those labels (the data are repeated)."""
        kw = {**(dim if isinstance(dim, dict) else {}), **kw}
        out = self
        for name, labels in reversed(list(kw.items())):
            lab = np.asarray(labels.data if isinstance(labels, DataArray) else labels)
            data = np.broadcast_to(out.data[None], (lab.shape[0],) + out.data.shape).copy()
            out = DataArray(data, {**{k: v for k, v in out.coords.items()}, name: lab}, (name,) + out.dims, out.name, out.attrs)
        return out

    def dropna(self, dim, how="any"):
        ax = self.dims.index(dim)
        other = tuple(i for i in range(self.ndim) if i != ax)
        bad = np.isnan(self.data.astype(np.float64))
        drop = bad.any(axis=other) if how == "any" else bad.all(axis=other)
        return self.isel(**{dim: np.flatnonzero(~drop)})

    def squeeze2(self, dim=None):
        if dim is None:
            return squeeze(self)
        dims = [dim] if isinstance(dim, str) else list(dim)
        assert all(self.sizes[d] == 1 for d in dims)
        keep = [d for d in self.dims if d not in dims]
        # (the squeezed dimension's label stays behind as a scalar coordinate, as in xarray)
        out = DataArray(self.data.reshape([self.sizes[d] for d in keep]),
                        {k: (v if k in keep else np.asarray(v).reshape(())) for k, v in self.coords.items()}, keep,
                        self.name, self.attrs)
        return out

    def interp(self, coords=None, method="linear", kwargs=None, **kw):
        """1-D linear interpolation onto new labels of one dimension: xarray hands this to
        scipy.interpolate.interp1d (bounds_error=False, the caller's fill_value) on the index turned into floats
        (datetimes: nanoseconds from the smallest label) -- scipy itself is executed here."""
        from scipy.interpolate import interp1d

        coords = {**(coords or {}), **kw}
        assert len(coords) == 1 and method == "linear"
        (dim, new), = coords.items()
        newv = np.asarray(new.data if isinstance(new, DataArray) else new)
        x = np.asarray(self.coords[dim])
        if x.dtype.kind == "M":
            x0 = x.min()
            xf = (x - x0).astype("timedelta64[ns]").astype(np.float64)
            qf = (newv.astype(x.dtype) - x0).astype("timedelta64[ns]").astype(np.float64)
        else:
            xf, qf = x.astype(np.float64), newv.astype(np.float64)
        ax = self.dims.index(dim)
        f = interp1d(xf, self.data.astype(np.float64), kind="linear", axis=ax, bounds_error=False, copy=False,
                     assume_sorted=False, **(kwargs or {}))
        newdim = new.dims[0] if isinstance(new, DataArray) else dim
        dims = tuple(newdim if d == dim else d for d in self.dims)
        cs = {k: v for k, v in self.coords.items() if k != dim and np.ndim(v) == 1}
        cs[newdim] = newv
        if newdim != dim:
            cs[dim] = newv  # xarray keeps the interpolated dimension's labels as a coordinate on the new dimension
        out = DataArray(f(qf), cs, dims, self.name, self.attrs)
        return out

    def getattr_coord(self, name):
        if name.startswith("_") or name not in self.__dict__.get("coords", {}):
            raise AttributeError(name)
        return self[name]

    for f in (isnull, where, transpose, isel, drop_vars, to_dataset, sel, mean, fillna, _reduce, assign_coords,
              reindex, reindex_like, coarsen, pipe, assign_attrs, rename, equals, resample, diff, idxmin, sortby,
              expand_dims, dropna, interp):
        setattr(DataArray, f.__name__, f)
    DataArray.squeeze = squeeze2
    DataArray.__getattr__ = getattr_coord
    DataArray.min, DataArray.max = amin, amax
    DataArray.any = any_
    DataArray.all = lambda self: DataArray(np.all(self.data), dims=[])
    DataArray.__iter__ = iterate
    DataArray.size = property(lambda self: int(self.data.size))
    DataArray.chunks = None
    DataArray.loc = property(lambda self: _Loc(self))
    DataArray.__getitem__ = getitem
    DataArray.__setitem__ = setitem


_methods()


class _DsCoords(dict):
    def __setitem__(self, k, v):  # ds.coords[name] = (dim, values, attrs)
        if isinstance(v, tuple):
            v = v[1]
        super().__setitem__(k, np.asarray(v.data if isinstance(v, DataArray) else v))


class Dataset:
    def __init__(self, data_vars=None, coords=None, attrs=None):
        self._vars = {}
        self.coords = _DsCoords()
        self.attrs = dict(attrs or {})
        for k, v in (coords or {}).items():
            self.coords[k] = np.asarray(v.data if isinstance(v, DataArray) else v)
        for k, v in (data_vars or {}).items():
            self[k] = v

    def __setitem__(self, name, v):
        if isinstance(v, tuple):
            v = DataArray(v[1], dims=v[0])
        if not isinstance(v, DataArray):
            v = DataArray(v, dims=[] if np.ndim(v) == 0 else None)
        v = DataArray(v, name=name)
        for d, n in v.sizes.items():
            if d in self.coords:
                assert len(self.coords[d]) == n, (name, d)
            elif d in v.coords:
                self.coords[d] = v.coords[d]
        self._vars[name] = v

    def __getitem__(self, name):
        if name in self._vars:
            v = self._vars[name]
            return DataArray(v.data, {d: self.coords[d] for d in v.dims if d in self.coords}, v.dims, name, v.attrs)
        if name in self.coords:
            return DataArray(self.coords[name], {name: self.coords[name]}, [name], name)
        raise KeyError(name)

    def __contains__(self, name):
        return name in self._vars or name in self.coords

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError(name)
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name) from None

    @property
    def sizes(self):
        out = {k: len(v) for k, v in self.coords.items()}
        for v in self._vars.values():
            out.update(v.sizes)
        return out

    @property
    def data_vars(self):
        return self._vars

    def assign_attrs(self, attrs=None, **kw):
        out = Dataset(coords=self.coords, attrs={**self.attrs, **(attrs or {}), **kw})
        for k, v in self._vars.items():
            out[k] = v
        return out

    def copy(self):
        return self.assign_attrs()

    def assign_coords(self, coords=None, **kw):
        out = self.copy()
        for k, v in {**(coords or {}), **kw}.items():
            out.coords[k] = v  # (dim, values) tuples keep the values; the dimension is checked by swap_dims
        return out

    def swap_dims(self, mapping):
        """The coordinate ``new`` (1-D along ``old``) becomes the index of that dimension; ``old`` stays as a
        plain variable along the renamed dimension."""
        out = Dataset(attrs=self.attrs)
        for k, v in self.coords.items():
            out.coords[k] = v
        for old, new in mapping.items():
            assert len(self.coords[new]) == len(self.coords[old])
        for k, v in self._vars.items():
            out._vars[k] = DataArray(v.data, None, [mapping.get(d, d) for d in v.dims], k, v.attrs)
        for old, new in mapping.items():
            out._vars[old] = DataArray(out.coords.pop(old), None, [new], old)
        return out

    def pipe(self, f, *a, **k):
        return f(self, *a, **k)

    def drop_dims(self, names, errors="raise"):
        names = [names] if isinstance(names, str) else list(names)
        out = Dataset(coords={k: v for k, v in self.coords.items() if k not in names}, attrs=self.attrs)
        for k, v in self._vars.items():
            if not set(v.dims) & set(names):
                out[k] = v
        return out

    def transpose(self, *dims):
        out = Dataset(coords=self.coords, attrs=self.attrs)
        for k, v in self._vars.items():
            out[k] = v.transpose(*[d for d in dims if d in v.dims]) if set(v.dims) <= set(dims) else v
        return out

    def merge(self, other):
        out = Dataset(coords=self.coords, attrs=self.attrs)
        for k, v in self._vars.items():
            out[k] = v
        others = other.data_vars.items() if isinstance(other, Dataset) else [(other.name, other)]
        for k, v in others:
            out[k] = v
        return out


def where(cond, x, y):
    """xr.where for operands with the same SET of dims as the condition (or scalars)."""
    c = cond if isinstance(cond, DataArray) else DataArray(cond)

    def conform(o):
        if not isinstance(o, DataArray):
            return o
        assert set(o.dims) == set(c.dims), (o.dims, c.dims)
        return o.transpose(*c.dims).data

    xd, yd = conform(x), conform(y)
    return c._like(np.where(c.data, xd, yd))


def apply_ufunc(func, *das, input_core_dims=None, output_core_dims=None, vectorize=False, **kw):
    """xr.apply_ufunc(vectorize=True) for inputs with the SAME dims: core dims moved last, ``func`` called
    on every core-dim slab, output dims = loop dims + output core dims (ek80_complex.py:356-364,
    clean/api.py:256-264, 348-357)."""
    core = list(input_core_dims[0])
    assert vectorize and list(output_core_dims[0]) == core
    assert all(list(c) == core for c in input_core_dims) and all(set(d.dims) == set(das[0].dims) for d in das)
    da = das[0]
    loop = [d for d in da.dims if d not in core]
    arrs = [d.transpose(*(loop + core)).data for d in das]
    a, out = arrs[0], None
    for idx in np.ndindex(*a.shape[:len(loop)]):
        r = np.asarray(func(*[x[idx] for x in arrs]))
        if out is None:
            out = np.empty(a.shape, dtype=r.dtype)
        out[idx] = r
    return da._like(out, loop + core)


def concat(objs, dim):
    """xr.concat of same-shaped arrays along a NEW leading dimension (clean/utils.py:179, 313)."""
    first = objs[0]
    assert dim not in first.dims and all(o.dims == first.dims and o.shape == first.shape for o in objs)
    for o in objs[1:]:
        for d in first.dims:
            assert np.array_equal(o.coords[d], first.coords[d]), d
    return DataArray(np.stack([o.data for o in objs]), {k: v for k, v in first.coords.items() if np.ndim(v) == 1},
                     (dim,) + first.dims, first.name, first.attrs)


def full_like(da, fill, dtype=None):
    return da._like(np.full(da.shape, fill, dtype=dtype or da.dtype))


def zeros_like(da, dtype=None):
    return da._like(np.zeros(da.shape, dtype=dtype or da.dtype))


def merge(objs, **kw):
    out = Dataset()
    for o in objs:
        out = out.merge(o)
    return out
