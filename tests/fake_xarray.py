"""A duck-typed stand-in for the part of xarray's public API the drop-in's boundary touches (xarray itself cannot be
installed in this image: no network).  TEST INFRASTRUCTURE: constructor signatures, ``.values / .dims / .coords /
.attrs / .name / .data_vars``, item assignment with ``(dims, values[, attrs])`` tuples and ``assign_attrs`` behave
like xarray's; nothing else is offered, so anything else the boundary tried to use would fail loudly."""
import numpy as np


class DataArray:
    def __init__(self, data, dims=None, coords=None, attrs=None, name=None):
        self.values = np.asarray(data)
        self.dims = tuple(dims) if dims is not None else tuple(f"dim_{i}" for i in range(self.values.ndim))
        self.coords = {k: (v if isinstance(v, DataArray) else DataArray(np.asarray(v), (k,))) for k, v in (coords or {}).items()} \
            if self.values.ndim or coords else {}
        self.attrs = dict(attrs or {})
        self.name = name

    @property
    def ndim(self):
        return self.values.ndim

    @property
    def shape(self):
        return self.values.shape


class Dataset:
    def __init__(self, data_vars=None, coords=None, attrs=None):
        self.coords, self.data_vars, self.attrs = {}, {}, dict(attrs or {})
        for k, v in (coords or {}).items():
            dims, vals, at = (v + ({},))[:3] if isinstance(v, tuple) else ((k,), v, {})
            self.coords[k] = DataArray(vals, dims, attrs=at, name=k)
        for k, v in (data_vars or {}).items():
            self[k] = v

    def __setitem__(self, name, v):
        if isinstance(v, tuple):
            dims, vals, at = (v + ({},))[:3]
            v = DataArray(vals, dims, attrs=at)
        da = DataArray(v.values, v.dims, {d: self.coords[d] for d in v.dims if d in self.coords}, v.attrs, name)
        self.data_vars[name] = da

    def __getitem__(self, name):
        return self.data_vars[name] if name in self.data_vars else self.coords[name]

    def __contains__(self, name):
        return name in self.data_vars or name in self.coords

    def assign_attrs(self, attrs=None, **kw):
        out = Dataset(attrs={**self.attrs, **(attrs or {}), **kw})
        out.coords, out.data_vars = self.coords, self.data_vars
        return out


# ---- a DataTree-like container, as echopype's EchoData wraps one (echodata/echodata.py:43-346) -----------------------
class DataTreeNode:
    """One group of the tree: ``.ds`` / ``.dataset`` / ``.to_dataset()`` give its Dataset, ``.children`` its sub-groups."""

    def __init__(self, ds=None, children=None):
        self._ds = ds if ds is not None else Dataset()
        self.children = dict(children or {})

    @property
    def ds(self):
        return self._ds

    dataset = ds

    def to_dataset(self):
        return self._ds

    def __getitem__(self, path):
        node = self
        for part in path.split("/"):
            node = node.children[part]
        return node


class TreeEchoData:
    """The read API of echopype's EchoData over a DataTree: ``ed[path]`` walks the tree and returns the group's Dataset,
    or None for a group the file does not have (echodata.py:327-335); ``group_paths`` lists what exists;
    ``sonar_model`` / ``source_file`` / ``converted_raw_path`` as attributes."""

    def __init__(self, sonar_model, root, source_file=None):
        self.sonar_model, self.source_file, self.converted_raw_path = sonar_model, source_file, None
        self._tree = root

    @property
    def group_paths(self):
        out = ["Top-level"]

        def walk(node, prefix):
            for name, child in node.children.items():
                out.append(prefix + name)
                walk(child, prefix + name + "/")

        walk(self._tree, "")
        return out

    def __getitem__(self, key):
        if key in (None, "Top-level"):
            return self._tree.ds
        try:
            return self._tree[key].ds
        except KeyError:
            return None
