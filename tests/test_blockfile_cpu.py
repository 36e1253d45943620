"""On-disk block output: pyqmc_amd.blockfile against the layout the REFERENCE writes (tests/golden/
g27_hdf_layout.npz: names, shapes and dtype kinds of every dataset of the reference's vmc / rundmc files, recorded by
make_golden.py::g_hdf_layout through an in-memory h5py stand-in).  The image has no HDF5 library, so the NumPy-archive back end
is what runs here; the GPU twin (tests/test_gpu_parity.py::test_vmc_and_dmc_write_the_reference_layout) drives it from the
real vmc / rundmc loops."""

import json

import numpy as np
import pytest

from helpers import golden
from pyqmc_amd import blockfile
from pyqmc_amd.configs import OpenConfigs, PeriodicConfigs


def ref_layout():
    return json.loads(str(golden("g27_hdf_layout")["layout"]))


def fake_block(layout, i, rng):
    blk = {}
    for k, (shape, kind) in layout.items():
        if k in blockfile.STATE_KEYS:
            continue
        blk[k] = i if kind == "i" else float(rng.standard_normal())
    return blk


@pytest.mark.parametrize("which", ["vmc", "dmc"])
def test_npz_store_reproduces_the_reference_layout(tmp_path, which):
    lay = ref_layout()[which]
    nb = lay["block"][0][0]
    rng = np.random.default_rng(0)
    cfg = OpenConfigs(rng.standard_normal(lay["configs"][0]))
    w = np.ones(lay["configs"][0][0]) if which == "dmc" else None
    f = blockfile.BlockFile(str(tmp_path / "run.hdf5"), backend="npz")
    assert not f.exists() and f.last_block() is None
    blocks = [fake_block(lay, i, rng) for i in range(nb)]
    for b in blocks:
        cfg.configs += 0.1
        f.append(b, {"tstep": 0.3} if which == "vmc" else {}, cfg, w)
    got = {k: [list(s), kind] for k, (s, kind) in f.listing().items()}
    assert got == lay  # same dataset names, shapes (blocks first) and dtype kinds as the reference's file
    assert sorted(f.attrs()) == ref_layout()[which + "_attrs"]
    assert f.last_block() == nb - 1 and np.array_equal(f.datasets()["block"], np.arange(nb))
    # restart state = the walkers of the last block
    new = OpenConfigs(np.zeros_like(cfg.configs))
    wts = f.load_walkers(new)
    assert np.array_equal(new.configs, cfg.configs) and ((wts is None) if which == "vmc" else np.array_equal(wts, w))
    out = blockfile.read_mc_output(str(tmp_path / "run.hdf5"), warmup=1)
    vals = np.array([b["energytotal"] for b in blocks[1:]])
    assert abs(out["energytotal"] - vals.mean()) < 1e-15 and "configs" not in out and "block" not in out
    if len(vals) > 1:
        assert abs(out["energytotal_err"] - vals.std(ddof=1) / np.sqrt(len(vals))) < 1e-15


def test_periodic_walkers_keep_wrap_counters_and_h5py_absence_is_loud(tmp_path):
    lat = np.eye(3) * 5.0
    cfg = PeriodicConfigs(np.random.default_rng(1).random((4, 2, 3)) * 12 - 3, lat)
    f = blockfile.BlockFile(str(tmp_path / "p"), backend="npz")
    f.append({"block": 0, "energytotal": 1.0}, {}, cfg)
    back = PeriodicConfigs(np.zeros((4, 2, 3)), lat)
    f.load_walkers(back)
    assert np.array_equal(back.configs, cfg.configs) and np.array_equal(back.wrap, cfg.wrap) and np.abs(cfg.wrap).sum() > 0
    if blockfile.h5py is None:
        with pytest.raises(RuntimeError, match="h5py"):
            blockfile.BlockFile(str(tmp_path / "q"), backend="h5py")
        with pytest.raises(RuntimeError, match="h5py"):
            blockfile.to_hdf5(str(tmp_path / "p"), str(tmp_path / "p.h5"))


def test_h5py_backend_round_trip_matches_the_npz_store(tmp_path):
    """Runs wherever h5py exists (NOT in this image: the HDF5 branch of blockfile.py has never executed here — INTEGRATION.md
    says so).  The same blocks through both back ends give the same datasets, attributes and restart state; the converter
    reproduces the direct file; and the file has the reference's extendable-dataset layout (hdftools.py:19-53)."""
    h5py = pytest.importorskip("h5py")
    lay = ref_layout()["dmc"]
    rng = np.random.default_rng(0)
    cfg = OpenConfigs(rng.standard_normal(lay["configs"][0]))
    w = rng.random(lay["configs"][0][0])
    a = blockfile.BlockFile(str(tmp_path / "a.hdf5"), backend="h5py")
    b = blockfile.BlockFile(str(tmp_path / "b"), backend="npz")
    for i in range(4):
        blk = fake_block(lay, i, rng)
        cfg.configs += 0.1
        for f in (a, b):
            f.append(blk, {"tstep": 0.02}, cfg, w)
    da, db = a.datasets(with_state=True), b.datasets(with_state=True)
    assert sorted(da) == sorted(db)
    for k in da:
        assert np.array_equal(da[k], db[k]), k
    assert a.last_block() == b.last_block() == 3 and dict(a.attrs()).keys() == dict(b.attrs()).keys()
    with h5py.File(str(tmp_path / "a.hdf5"), "r") as f:
        assert f["energytotal"].maxshape[0] is None and f["energytotal"].shape == (4,)
    blockfile.to_hdf5(str(tmp_path / "b"), str(tmp_path / "c.hdf5"))
    dc = blockfile.BlockFile(str(tmp_path / "c.hdf5"), backend="h5py").datasets(with_state=True)
    for k in da:
        assert np.array_equal(da[k], dc[k]), k
    out_a, out_b = blockfile.read_mc_output(str(tmp_path / "a.hdf5")), blockfile.read_mc_output(str(tmp_path / "b"))
    assert out_a["energytotal"] == out_b["energytotal"]
