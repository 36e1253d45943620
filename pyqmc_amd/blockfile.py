"""On-disk block output — SURVEY.md §8(f4); the layout of ``pyqmc/method/hdftools.py:19-53`` as written by ``vmc_file``
(``mc.py:92-99``) and ``dmc_file`` (``dmc.py:379-391``) and read back by ``recipes.read_mc_output`` (``recipes.py:224-239``):

* one extendable dataset per block quantity, named exactly as the block dictionary's keys (``energytotal``, ``acceptance``,
  ``block``, ``nconfig``, ...; DMC adds ``weight``, ``e_trial``, ...), first axis = block;
* file attributes from the run (``tstep``, ...);
* the CURRENT walkers: ``configs`` (nconf, nelec, 3), periodic runs also ``wrap``, DMC also ``weights`` — overwritten every
  block (the restart state: a run continues from ``block[-1] + 1``, mc.py:235-243).

Two back ends behind one interface.  With ``h5py`` importable the file IS that HDF5 file — the reference's
``read_mc_output`` / ``Configs.load_hdf`` read it and files written by the reference's ``hdftools`` are read and continued here
(``tools/verify_hdf5_with_reference.py``, real h5py 3.3.0 / HDF5 1.10.6: ``profiles/r04_f4_reference_hdf5.txt``).  The interpreter
the GPU stack runs under has no h5py, so otherwise the same layout goes into two NumPy archives — ``<name>.blocks.npz`` (an
append-only zip: member ``<dataset>/<block index>.npy``) and ``<name>.state.npz`` (configs / wrap / weights, replaced
atomically) — and ``to_hdf5`` converts them: in-process where h5py exists, else through any interpreter that has it
(``PQA_H5PY_PYTHON``; this image: ``/opt/conda/bin/python3.9``).  ``read_mc_output`` here reads either form.  One deliberate
difference from the reference's file: the walkers are stored as float64 (the reference's ``initialize_hdf`` names no dtype and
h5py then keeps float32: its restarts round the walkers; ``load_hdf`` accepts either).
"""

import io
import os
import zipfile

import numpy as np

try:  # optional: not in this image
    import h5py
except ImportError:
    h5py = None

STATE_KEYS = ("configs", "wrap", "weights")


def _is_npz_store(path):
    return os.path.exists(path + ".blocks.npz") or os.path.exists(path + ".state.npz")


class BlockFile:
    """``BlockFile(path)``: ``append(block, attrs, configs, weights=None)`` after every block, ``last_block()`` and
    ``load_walkers(configs)`` to continue a run, ``datasets()`` -> {name: array with the block axis first}."""

    def __init__(self, path, backend=None):
        self.path = path
        self._nblocks = None  # npz store: records written so far (counted once, then kept: append is O(1) per block)
        self.backend = backend or ("h5py" if h5py is not None else "npz")
        if self.backend == "h5py" and h5py is None:
            raise RuntimeError("h5py is not installed: use backend='npz' and convert with blockfile.to_hdf5 where it is")

    # ---------------------------------------------------------------- existence / restart
    def exists(self):
        return os.path.isfile(self.path) if self.backend == "h5py" else _is_npz_store(self.path)

    def last_block(self):
        """Index of the last block written, or None."""
        if not self.exists():
            return None
        b = self.datasets().get("block")
        return None if b is None or len(b) == 0 else int(b[-1])

    def load_walkers(self, configs):
        """Restart state into ``configs`` (in place, like ``Configs.load_hdf`` coord.py:108-112, :248-252); returns the stored
        weights or None."""
        st = self._state()
        configs.configs = np.array(st["configs"], dtype=float)
        if "wrap" in st and hasattr(configs, "wrap"):
            configs.wrap = np.array(st["wrap"])
        return np.array(st["weights"]) if "weights" in st else None

    # ---------------------------------------------------------------- writing
    def append(self, block, attrs, configs, weights=None):
        state = {"configs": np.asarray(configs.configs)}
        if hasattr(configs, "wrap"):
            state["wrap"] = np.asarray(configs.wrap)
        if weights is not None:
            state["weights"] = np.asarray(weights)
        block = {k: np.asarray(v) for k, v in block.items()}
        if self.backend == "h5py":
            with h5py.File(self.path, "a") as f:
                for k, v in block.items():
                    if k not in f:
                        f.create_dataset(k, (0,) + v.shape, maxshape=(None,) + v.shape, dtype=v.dtype)
                    f[k].resize(f[k].shape[0] + 1, axis=0)
                    f[k][-1] = v
                for k, v in attrs.items():
                    f.attrs[k] = v
                for k, v in state.items():
                    if k in f and f[k].shape != v.shape:
                        f[k].resize(v.shape)
                    if k not in f:
                        f.create_dataset(k, v.shape, maxshape=(None,) + v.shape[1:], chunks=True, dtype=v.dtype)
                    f[k][...] = v
            return
        if self._nblocks is None:
            idx = self._block_index()
            self._nblocks = (idx[-1] + 1) if idx else 0
        n = self._nblocks
        # walkers first, record second: a crash in between leaves walkers one block AHEAD of the record (that block's averages
        # are lost), never a record whose walkers are one block behind
        tmp = self.path + ".state.tmp.npz"
        np.savez(tmp, **state)
        os.replace(tmp, self.path + ".state.npz")
        with zipfile.ZipFile(self.path + ".blocks.npz", "a", zipfile.ZIP_STORED) as z:
            for k, v in block.items():
                buf = io.BytesIO()
                np.save(buf, v)
                z.writestr(f"{k}/{n:08d}.npy", buf.getvalue())
            for k, v in attrs.items():
                buf = io.BytesIO()
                np.save(buf, np.asarray(v))
                z.writestr(f"__attrs__/{k}/{n:08d}.npy", buf.getvalue())
        self._nblocks = n + 1

    # ---------------------------------------------------------------- reading
    def _block_index(self):
        p = self.path + ".blocks.npz"
        if not os.path.exists(p):
            return []
        with zipfile.ZipFile(p) as z:
            return sorted({int(os.path.basename(n)[:-4]) for n in z.namelist() if not n.startswith("__attrs__/")})

    def _state(self):
        if self.backend == "h5py":
            with h5py.File(self.path, "r") as f:
                return {k: f[k][()] for k in STATE_KEYS if k in f}
        with np.load(self.path + ".state.npz") as z:
            return {k: z[k] for k in z.files}

    def attrs(self):
        if self.backend == "h5py":
            with h5py.File(self.path, "r") as f:
                return dict(f.attrs)
        out = {}
        with zipfile.ZipFile(self.path + ".blocks.npz") as z:
            for n in sorted(z.namelist()):
                if n.startswith("__attrs__/"):
                    out[n.split("/")[1]] = np.load(io.BytesIO(z.read(n)))[()]
        return out

    def datasets(self, with_state=False):
        """{dataset name: (nblocks, ...) array} — exactly the per-block datasets of the HDF5 layout."""
        if self.backend == "h5py":
            with h5py.File(self.path, "r") as f:
                return {k: f[k][()] for k in f if with_state or k not in STATE_KEYS}
        cols = {}
        with zipfile.ZipFile(self.path + ".blocks.npz") as z:
            for n in sorted(z.namelist()):
                if not n.startswith("__attrs__/"):
                    cols.setdefault(n.split("/")[0], []).append(np.load(io.BytesIO(z.read(n))))
        out = {k: np.stack(v) for k, v in cols.items()}
        if with_state:
            out.update(self._state())
        return out

    def listing(self):
        """{name: (shape, dtype kind)} of every dataset incl. the walker state — what the layout tests compare."""
        return {k: (tuple(v.shape), v.dtype.kind) for k, v in self.datasets(with_state=True).items()}


def h5py_interpreter():
    """An interpreter with h5py + numpy for ``to_hdf5`` where this one has none: ``$PQA_H5PY_PYTHON``, else the first ``python3`` /
    ``python`` on PATH (and conda's usual prefix) that imports both; None if none does."""
    import shutil
    import subprocess

    cands = [os.environ.get("PQA_H5PY_PYTHON")] + [shutil.which(n) for n in ("python3", "python")] + \
            [os.path.join(os.environ.get("CONDA_PREFIX", "/opt/conda"), "bin", "python")]
    for cand in cands:
        if cand and os.path.exists(cand):
            if subprocess.run([cand, "-c", "import h5py, numpy"], capture_output=True).returncode == 0:
                return cand
    return None


def to_hdf5(src, dst):
    """Convert an npz block store into the reference's HDF5 file: with h5py in-process, else by running this module's converter
    under an interpreter that has h5py (``h5py_interpreter``); RuntimeError if there is none."""
    if h5py is None:
        import subprocess

        py = h5py_interpreter()
        if py is None:
            raise RuntimeError("to_hdf5 needs h5py (none importable here, no interpreter with h5py found: set PQA_H5PY_PYTHON)")
        code = ("import importlib.util, sys; sp = importlib.util.spec_from_file_location('pqa_blockfile', sys.argv[1]); "
                "m = importlib.util.module_from_spec(sp); sp.loader.exec_module(m); m.to_hdf5(sys.argv[2], sys.argv[3])")
        r = subprocess.run([py, "-W", "ignore", "-c", code, os.path.abspath(__file__), src, dst], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("to_hdf5 under " + py + " failed: " + r.stderr[-500:])
        return
    store = BlockFile(src, backend="npz")
    with h5py.File(dst, "w") as f:
        for k, v in store.datasets().items():
            f.create_dataset(k, data=v, maxshape=(None,) + v.shape[1:])
        for k, v in store._state().items():
            f.create_dataset(k, data=v, maxshape=(None,) + v.shape[1:], chunks=True)
        for k, v in store.attrs().items():
            f.attrs[k] = v


def read_mc_output(fname, warmup=1, reblock=None, exclude_keys=("configs", "weights", "block", "nconfig", "wrap")):
    """Means and standard errors over blocks (``recipes.read_mc_output`` recipes.py:224-239) of an HDF5 file or an npz store."""
    store = BlockFile(fname, backend="npz" if _is_npz_store(fname) else "h5py")
    ret = {"fname": fname, "warmup": warmup, "reblock": reblock}
    for k, vals in store.datasets().items():
        if k in exclude_keys:
            continue
        vals = vals[warmup:]
        if reblock is not None:  # reblock.reblock: means of `reblock` equal chunks (remainder dropped from the front)
            n = (len(vals) // reblock) * reblock
            vals = vals[len(vals) - n :].reshape((reblock, n // reblock) + vals.shape[1:]).mean(axis=1)
        ret[k] = np.mean(vals, axis=0)
        ret[k + "_err"] = np.std(vals, axis=0, ddof=1) / np.sqrt(len(vals)) if len(vals) > 1 else np.full(np.shape(ret[k]), np.nan)
    return ret
