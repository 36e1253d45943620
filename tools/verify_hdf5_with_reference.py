"""f4 cross-check against the REFERENCE with a real HDF5 library (run in the build container, where /root/reference exists, under an
interpreter that has h5py — this image: /opt/conda/bin/python3.9, h5py 3.3.0 / HDF5 1.10.6):

    /opt/conda/bin/python3.9 tools/verify_hdf5_with_reference.py > profiles/r04_f4_reference_hdf5.txt

1. a block file written by pyqmc_amd.blockfile's h5py branch is read by the reference's recipes.read_mc_output (recipes.py:224-239)
   and gives what our read_mc_output gives (same keys, means, standard errors), with and without reblocking;
2. a file written by the reference's hdftools.setup_hdf / append_hdf (hdftools.py:19-53) + OpenConfigs.to_hdf (coord.py:98-106) is
   read by pyqmc_amd.blockfile: datasets, attributes, restart state, last block;
3. a run CONTINUED by our BlockFile on the reference-written file stays readable by the reference.
pyscf is mocked (not in the image), numba becomes an identity decorator; h5py, numpy, scipy, pandas are real."""

import importlib.util
import os
import sys
import tempfile
import types
from unittest import mock

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("PQA_REFERENCE", "/root/reference")
nb = types.ModuleType("numba")
nb.njit = nb.jit = lambda *a, **k: (a[0] if a and callable(a[0]) else (lambda f: f))
nb.prange = range
sys.modules["numba"] = nb
for m in ("pyscf", "pyscf.gto", "pyscf.scf", "pyscf.pbc", "pyscf.pbc.gto", "pyscf.pbc.scf", "pyscf.lib", "pyscf.lib.chkfile", "pyscf.mcscf",
          "pyscf.pbc.dft", "pyscf.pbc.dft.gen_grid", "pyscf.pbc.gto.cell", "pyscf.pbc.lib", "pyscf.pbc.lib.kpts_helper", "pyscf.fci", "pyscf.fci.addons",
          "pyscf.pbc.tools", "pyscf.pbc.tools.pbc", "pyscf.pbc.gto.eval_gto", "pyscf.hci", "pyscf.tdscf", "pyscf.ci", "pyscf.cc"):
    sys.modules.setdefault(m, mock.MagicMock(name=m))
sys.path.insert(0, REF)
import h5py  # noqa: E402
import pyqmc.method.hdftools as hdftools  # noqa: E402
import pyqmc.recipes as recipes  # noqa: E402
from pyqmc.configurations.coord import OpenConfigs as RefOpenConfigs  # noqa: E402


def load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "pyqmc_amd", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


blockfile = load("blockfile")
configs = load("configs")
assert blockfile.h5py is h5py
print(f"h5py {h5py.__version__}, HDF5 {h5py.version.hdf5_version}, numpy {np.__version__}; reference at {REF}")
rng = np.random.default_rng(4)
keys_f = ("energytotal", "energyke", "energyee", "energyei", "energyecp", "energygrad2", "acceptance")
tmp = tempfile.mkdtemp()


def block(i):
    b = {k: float(rng.standard_normal()) for k in keys_f}
    b.update(block=i, nconfig=6, obdmvalue=rng.standard_normal((3, 3)), cplx=complex(rng.standard_normal(), rng.standard_normal()))
    return b


def same(a, b, what):
    keys = sorted(k for k in a if k not in ("fname",))
    assert keys == sorted(k for k in b if k not in ("fname",)), (what, keys, sorted(b))
    worst = 0.0
    for k in keys:
        if a[k] is None or isinstance(a[k], (int, str)):
            assert a[k] == b[k], (what, k)
            continue
        d = np.max(np.abs(np.asarray(a[k]) - np.asarray(b[k])))
        assert d < 1e-13, (what, k, d)
        worst = max(worst, float(d))
    print(f"  {what}: {len(keys)} entries equal (max abs difference {worst:.1e})")


# 1. ours -> reference
p1 = os.path.join(tmp, "ours.hdf5")
bf = blockfile.BlockFile(p1, backend="h5py")
cfg = configs.OpenConfigs(rng.standard_normal((6, 8, 3)))
for i in range(9):
    cfg.configs += 0.05
    bf.append(block(i), {"tstep": 0.3}, cfg)
print("1. written by pyqmc_amd.blockfile (h5py branch), read by the reference's recipes.read_mc_output")
for kw in (dict(warmup=1), dict(warmup=2, reblock=None), dict(warmup=1, reblock=4)):
    same(recipes.read_mc_output(p1, **kw), blockfile.read_mc_output(p1, **kw), f"read_mc_output({kw})")
with h5py.File(p1, "r") as f:
    rc = RefOpenConfigs(np.zeros((6, 8, 3)))
    rc.load_hdf(f)  # coord.py:108-112: the reference's restart read
    assert np.array_equal(rc.configs, cfg.configs) and f.attrs["tstep"] == 0.3 and f["energytotal"].maxshape == (None,)
print("  the reference's Configs.load_hdf restores our walkers; attrs and extendable datasets as hdftools makes them")

# 2. reference -> ours
p2 = os.path.join(tmp, "ref.hdf5")
rcfg = RefOpenConfigs(rng.standard_normal((6, 8, 3)))
blocks = [block(i) for i in range(5)]
with h5py.File(p2, "a") as f:
    hdftools.setup_hdf(f, blocks[0], {"tstep": 0.25})
    rcfg.initialize_hdf(f)  # mc.py:92-99 vmc_file
for b in blocks:
    rcfg.configs += 0.01
    with h5py.File(p2, "a") as f:
        hdftools.append_hdf(f, b)
        rcfg.to_hdf(f)
st = blockfile.BlockFile(p2, backend="h5py")
ds = st.datasets()
for k in blocks[0]:
    assert np.array_equal(ds[k], np.array([b[k] for b in blocks])), k
mine = configs.OpenConfigs(np.zeros((1, 1, 3)))
# (the reference's initialize_hdf gives no dtype, so h5py stores ITS walkers in single precision; ours are written as float64)
assert st.load_walkers(mine) is None and np.array_equal(mine.configs, rcfg.configs.astype(np.float32).astype(float))
assert st.last_block() == 4 and float(st.attrs()["tstep"]) == 0.25
print(f"2. written by the reference's hdftools + Configs.to_hdf, read by pyqmc_amd.blockfile: {len(ds)} datasets equal, restart state and attrs equal, last block 4")
same(recipes.read_mc_output(p2), blockfile.read_mc_output(p2), "read_mc_output on the reference's file")

# 3. ours continues the reference's file
for i in range(5, 8):
    mine.configs += 0.02
    st.append(block(i), {"tstep": 0.25}, mine)
out = recipes.read_mc_output(p2, warmup=0)
with h5py.File(p2, "r") as f:
    assert f["block"][()].tolist() == list(range(8)) and np.array_equal(f["configs"][()], mine.configs.astype(np.float32))
same(out, blockfile.read_mc_output(p2, warmup=0), "read_mc_output after our continuation of the reference's file")
print("3. a run continued by pyqmc_amd.blockfile on the reference-written file stays readable by the reference (8 blocks)")
print("ok")
