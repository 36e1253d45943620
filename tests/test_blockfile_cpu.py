"""On-disk block output: pyqmc_amd.blockfile against the layout the REFERENCE writes (tests/golden/
g27_hdf_layout.npz: names, shapes and dtype kinds of every dataset of the reference's vmc / rundmc files, recorded by
make_golden.py::g_hdf_layout through an in-memory h5py stand-in).  The image has no HDF5 library, so the NumPy-archive back end
is what runs here; the GPU twin (tests/test_gpu_parity.py::test_vmc_and_dmc_write_the_reference_layout) drives it from the
real vmc / rundmc loops."""

import json
import os

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
        py = blockfile.h5py_interpreter()
        if py is None:
            with pytest.raises(RuntimeError, match="h5py"):
                blockfile.to_hdf5(str(tmp_path / "p"), str(tmp_path / "p.h5"))
        else:  # the converter runs under the interpreter that has h5py: a real HDF5 file with the same datasets and walkers
            import subprocess

            blockfile.to_hdf5(str(tmp_path / "p"), str(tmp_path / "p.h5"))
            assert open(tmp_path / "p.h5", "rb").read(8) == b"\x89HDF\r\n\x1a\n"
            code = ("import h5py, json, sys; f = h5py.File(sys.argv[1], 'r'); "
                    "print(json.dumps({k: [list(f[k].shape), f[k].dtype.kind, f[k][()].tolist(), f[k].maxshape[0] is None] for k in f}))")
            got = json.loads(subprocess.run([py, "-W", "ignore", "-c", code, str(tmp_path / "p.h5")], capture_output=True, text=True, check=True).stdout)
            assert sorted(got) == ["block", "configs", "energytotal", "wrap"] and all(v[3] for v in got.values())
            assert np.array_equal(np.array(got["configs"][2]), cfg.configs) and np.array_equal(np.array(got["wrap"][2]), cfg.wrap)
            assert got["energytotal"][:3] == [[1], "f", [1.0]] and got["block"][:3] == [[1], "i", [0]]


@pytest.fixture(params=["h5py", "stand-in"])
def h5(request, monkeypatch):
    """The HDF5 back end under test: the real h5py where it exists, and ALWAYS the in-memory stand-in (tests/fake_h5py.py),
    which is how the h5py branches of blockfile.py execute in images without an HDF5 library."""
    if request.param == "h5py":
        yield pytest.importorskip("h5py")
        return
    import fake_h5py

    monkeypatch.setattr(blockfile, "h5py", fake_h5py)
    yield fake_h5py
    fake_h5py.forget()


def test_h5py_backend_round_trip_matches_the_npz_store(tmp_path, h5):
    """The same blocks through both back ends give the same datasets, attributes and restart state; the converter reproduces
    the direct file; and the file has the reference's extendable-dataset layout (hdftools.py:19-53)."""
    lay = ref_layout()["dmc"]
    rng = np.random.default_rng(0)
    cfg = OpenConfigs(rng.standard_normal(lay["configs"][0]))
    w = rng.random(lay["configs"][0][0])
    a = blockfile.BlockFile(str(tmp_path / "a.hdf5"), backend="h5py")
    b = blockfile.BlockFile(str(tmp_path / "b"), backend="npz")
    assert not a.exists() and a.last_block() is None
    for i in range(4):
        blk = fake_block(lay, i, rng)
        cfg.configs += 0.1
        for f in (a, b):
            f.append(blk, {"tstep": 0.02}, cfg, w)
    da, db = a.datasets(with_state=True), b.datasets(with_state=True)
    assert sorted(da) == sorted(db) and a.listing() == b.listing()
    for k in da:
        assert np.array_equal(da[k], db[k]), k
    assert a.exists() and a.last_block() == b.last_block() == 3 and dict(a.attrs()).keys() == dict(b.attrs()).keys()
    with h5.File(str(tmp_path / "a.hdf5"), "r") as f:
        assert f["energytotal"].maxshape[0] is None and f["energytotal"].shape == (4,)
        assert f["configs"].maxshape[0] is None and f["configs"].shape == tuple(lay["configs"][0])
    # restart state: the walkers of the LAST block, weights included (mc.py:235-243, dmc.py:466-500)
    back = OpenConfigs(np.zeros((1, 1, 3)))
    wb = a.load_walkers(back)
    assert np.array_equal(back.configs, cfg.configs) and np.array_equal(wb, w)
    # a walker count that changes between blocks (a DMC restart with another population) resizes the state datasets
    cfg2 = OpenConfigs(rng.standard_normal((lay["configs"][0][0] + 3,) + tuple(lay["configs"][0][1:])))
    a.append(fake_block(lay, 4, rng), {"tstep": 0.02}, cfg2, np.ones(len(cfg2.configs)))
    assert a._state()["configs"].shape == cfg2.configs.shape and a.last_block() == 4
    blockfile.to_hdf5(str(tmp_path / "b"), str(tmp_path / "c.hdf5"))
    dc = blockfile.BlockFile(str(tmp_path / "c.hdf5"), backend="h5py").datasets(with_state=True)
    for k in db:
        assert np.array_equal(db[k], dc[k]), k
    out_a, out_b = blockfile.read_mc_output(str(tmp_path / "c.hdf5")), blockfile.read_mc_output(str(tmp_path / "b"))
    assert out_a["energytotal"] == out_b["energytotal"] and out_a["energytotal_err"] == out_b["energytotal_err"]


def test_vmc_and_rundmc_write_through_the_h5py_backend(tmp_path, h5):
    """The drivers' own file handling on the HDF5 back end: vmc(hdf_file=) appends one record per block and restarts from
    block[-1] + 1; rundmc(hdf_file=) stores weights and the per-block e_trial / e_est and continues from them — the same
    checks test_dmc_cpu makes on the npz store.  (Oracle wave function, protocol-route workers: no GPU.)"""
    import helpers
    import pyqmc_amd as pa
    from pyqmc_amd import dmc, systems
    from test_dmc_cpu import OracleAccumulator

    mol = systems.water()
    wf = helpers.oracle_wf(mol, systems.random_mf(mol))
    np.random.seed(1)
    cfg = pa.initial_guess(mol, 5, rng=np.random.default_rng(1))
    path = str(tmp_path / "v.hdf5")
    acc = {"energy": OracleAccumulator(mol)}
    df, cfg = pa.vmc(wf, cfg, nblocks=2, nsteps_per_block=1, tstep=0.3, accumulators=acc, hdf_file=path, worker=helpers.protocol_vmc_worker)
    st = blockfile.BlockFile(path)
    assert st.backend == "h5py" and st.datasets()["block"].tolist() == [0, 1] and float(st.attrs()["tstep"]) == 0.3
    df2, cfg2 = pa.vmc(wf, pa.initial_guess(mol, 5, rng=np.random.default_rng(9)), nblocks=3, nsteps_per_block=1, tstep=0.3, accumulators=acc,
                       hdf_file=path, worker=helpers.protocol_vmc_worker)
    assert df2["block"].tolist() == [2] and st.datasets()["block"].tolist() == [0, 1, 2]
    assert np.array_equal(st._state()["configs"], cfg2.configs)
    lay = ref_layout()["vmc"]
    assert {k: v[1] for k, v in st.listing().items()} == {k: v[1] for k, v in lay.items()}  # the reference's dataset names and dtype kinds
    dpath = str(tmp_path / "d.hdf5")
    kw = dict(tstep=0.05, nsteps_per_block=1, vmc_warmup=1, accumulators=acc, hdf_file=dpath, propagate=helpers.protocol_dmc_propagate,
              vmc_worker=helpers.protocol_vmc_worker)
    d1, c1, w1 = dmc.rundmc(wf, pa.initial_guess(mol, 5, rng=np.random.default_rng(2)), nblocks=2, **kw)
    d2, c2, w2 = dmc.rundmc(wf, pa.initial_guess(mol, 5, rng=np.random.default_rng(3)), nblocks=3, **kw)
    ds = blockfile.BlockFile(dpath)
    assert ds.datasets()["block"].tolist() == [0, 1, 2] and d2["block"].tolist() == [2] and d2["e_trial"][0] == d1["e_trial"][-1]
    assert np.array_equal(ds._state()["weights"], w2) and np.array_equal(ds._state()["configs"], c2.configs)
    assert sorted(ds.listing()) == sorted(ref_layout()["dmc"])


def test_chkfile_h5py_branch_on_the_reference_checkpoint(monkeypatch):
    """chkfile.py's h5py code path (f["mol"][()], f["scf"] group access, k-point lists) executed through the stand-in serving the
    reference's real file, against the built-in parser's result."""
    import fake_h5py
    from pyqmc_amd import chkfile

    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "files", "diamond_primitive.hdf5")
    monkeypatch.setattr(chkfile, "h5py", fake_h5py)
    assert chkfile.read_mol_json(path, "h5py") == chkfile.read_mol_json(path, "lite")
    (m1, f1), (m2, f2) = chkfile.load_scf(path, backend="h5py"), chkfile.load_scf(path, backend="lite")
    assert np.array_equal(f1.kpts, f2.kpts) and f1.e_tot == f2.e_tot and m1.nelec == m2.nelec
    for s in (0, 1):
        for k in range(8):
            assert np.array_equal(f1.mo_coeff[s][k], f2.mo_coeff[s][k]) and np.array_equal(f1.mo_occ[s][k], f2.mo_occ[s][k])


def h5py_interpreter():
    """An interpreter that has h5py, numpy and pytest (this image: the Anaconda python under /opt/conda; PQA_H5PY_PYTHON names another)."""
    import shutil
    import subprocess

    for cand in (os.environ.get("PQA_H5PY_PYTHON"), "/opt/conda/bin/python3.9", shutil.which("python3.9")):
        if cand and os.path.exists(cand):
            r = subprocess.run([cand, "-c", "import h5py, numpy, pytest, scipy"], capture_output=True)
            if r.returncode == 0:
                return cand
    return None


@pytest.mark.skipif(blockfile.h5py is not None, reason="h5py is importable here: the [h5py] cases above already ran")
def test_h5py_branches_against_a_real_hdf5_library_in_another_interpreter():
    """This interpreter has no h5py, the image's Anaconda python has (h5py 3.3.0 / HDF5 1.10.6): the [h5py] cases of this file
    (block round trip, restart, resize, converter, read_mc_output, vmc / rundmc writing and continuing their files) and the
    chkfile tests (h5py back end and the built-in parser against the real library on the reference's checkpoints) run THERE.
    tools/verify_hdf5_with_reference.py (profiles/r04_f4_reference_hdf5.txt) is the same against the reference's own
    read_mc_output / hdftools / Configs.to_hdf."""
    import subprocess

    py = h5py_interpreter()
    if py is None:
        pytest.skip("no interpreter with h5py in this image")
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([py, "-m", "pytest", "-q", "-p", "no:cacheprovider", "-rA", "-W", "ignore", os.path.join(here, "test_blockfile_cpu.py"),
                        os.path.join(here, "test_chkfile_cpu.py"), "-k", "h5py"], capture_output=True, text=True, cwd=os.path.dirname(here), timeout=600)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-3000:]
    for name in ("test_h5py_backend_round_trip_matches_the_npz_store[h5py]", "test_vmc_and_rundmc_write_through_the_h5py_backend[h5py]",
                 "test_scan_and_h5py_backends_agree_and_errors_are_loud"):
        assert f"PASSED tests/test_blockfile_cpu.py::{name}" in out or f"PASSED tests/test_chkfile_cpu.py::{name}" in out, out[-3000:]
