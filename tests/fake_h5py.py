"""TEST INFRASTRUCTURE — an in-memory stand-in for the part of the h5py API that pyqmc_amd.blockfile and pyqmc_amd.chkfile call
(neither image has an HDF5 library, so their ``h5py`` branches could otherwise never execute here).  It keeps h5py's rules for
that subset — a dataset is fixed-shape unless ``maxshape`` allows growing along an axis, ``resize`` beyond ``maxshape`` and
shape-mismatched assignments raise, modes ``r`` / ``a`` / ``w`` — so code that breaks them fails here as it would on a real
file.  A path that is an actual HDF5 file on disk (the reference's checkpoint files) is served read-only through
``pyqmc_amd.hdf5lite``.  tests/golden/make_golden.py records the reference's own writes through a cruder version (g27)."""

import os

import numpy as np

_STORE = {}


class Dataset:
    def __init__(self, shape, maxshape, dtype, data=None):
        arr = np.zeros(shape, dtype=dtype) if data is None else np.array(data, dtype=dtype)
        self._a = arr
        self.maxshape = tuple(arr.shape) if maxshape is None else tuple(maxshape)
        if len(self.maxshape) != arr.ndim or any(m is not None and m < s for m, s in zip(self.maxshape, arr.shape)):
            raise ValueError("maxshape incompatible with shape")

    shape = property(lambda self: self._a.shape)
    dtype = property(lambda self: self._a.dtype)

    def resize(self, size, axis=None):
        new = tuple(size) if axis is None else self._a.shape[:axis] + (int(size),) + self._a.shape[axis + 1 :]
        if len(new) != self._a.ndim:
            raise TypeError("resize: rank mismatch")
        for n, m, s in zip(new, self.maxshape, self._a.shape):
            if n != s and m is not None and n > m:
                raise ValueError(f"unable to set extent dims (new dimension {n} is larger than maxshape {m})")
        out = np.zeros(new, dtype=self._a.dtype)
        sl = tuple(slice(0, min(o, n)) for o, n in zip(self._a.shape, new))
        out[sl] = self._a[sl]
        self._a = out

    def __setitem__(self, key, value):
        self._a[key] = value  # numpy raises on shapes that do not broadcast, as h5py does

    def __getitem__(self, key):
        out = self._a[key]
        return out.copy() if isinstance(out, np.ndarray) else out

    def __len__(self):
        return self._a.shape[0]

    def __array__(self, dtype=None, copy=None):
        return np.array(self._a, dtype=dtype)


class _LiteNode:
    """Read-only view of a group / dataset of a real HDF5 file (through hdf5lite)."""

    def __init__(self, f, path):
        self._f, self._p = f, path

    def __getitem__(self, key):
        if key == () or key is Ellipsis:
            v = self._f[self._p]
            return v.encode() if isinstance(v, str) else v  # h5py returns variable-length strings as bytes
        q = self._p.rstrip("/") + "/" + key
        if q not in self._f:
            raise KeyError(key)
        return _LiteNode(self._f, q)

    def __contains__(self, key):
        return (self._p.rstrip("/") + "/" + key) in self._f

    def __iter__(self):
        return iter(self._f.keys(self._p))

    def __array__(self, dtype=None, copy=None):
        return np.array(self._f[self._p], dtype=dtype)


class File:
    def __init__(self, path, mode="r"):
        self._path, self._mode = path, mode
        self._lite = None
        if os.path.isfile(path) and path not in _STORE:  # a real file on disk
            if mode != "r":
                raise OSError("the stand-in opens real HDF5 files read-only")
            from pyqmc_amd.hdf5lite import File as Lite

            self._lite = _LiteNode(Lite(path), "/")
            return
        if mode == "r" and path not in _STORE:
            raise FileNotFoundError(path)
        if mode == "w" or path not in _STORE:
            _STORE[path] = {"data": {}, "attrs": {}}
            open(path, "wb").close()  # so that os.path.isfile() sees it, like a real file
        self._d = _STORE[path]["data"]
        self.attrs = _STORE[path]["attrs"]

    def _writable(self):
        if self._mode == "r":
            raise OSError("file opened read-only")

    def create_dataset(self, name, shape=None, maxshape=None, dtype=None, chunks=None, data=None):
        self._writable()
        if name in self._d:
            raise ValueError(f"unable to create dataset (name already exists: {name})")
        if maxshape is not None and any(m is None for m in maxshape) and chunks is False:
            raise ValueError("extendable datasets must be chunked")
        self._d[name] = Dataset(shape if data is None else np.shape(data), maxshape, dtype if dtype is not None else (np.asarray(data).dtype if data is not None else float), data)
        return self._d[name]

    def __getitem__(self, name):
        return self._lite[name] if self._lite is not None else self._d[name]

    def __contains__(self, name):
        return (name in self._lite) if self._lite is not None else (name in self._d)

    def __iter__(self):
        return iter(self._lite) if self._lite is not None else iter(sorted(self._d))

    def keys(self):
        return list(iter(self))

    def close(self):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def forget(path=None):
    """Drop one stored file (or all of them)."""
    if path is None:
        for p in list(_STORE):
            forget(p)
        return
    _STORE.pop(path, None)
    if os.path.isfile(path) and os.path.getsize(path) == 0:
        os.remove(path)
