"""PySCF checkpoint ingest (second half): pyqmc_amd.chkfile on the reference's OWN checkpoint files
(tests/golden/files/*.hdf5 — data files of the reference's test suite, /root/reference/tests/files) against what the REFERENCE
builds from the same files (tests/golden/g28_chk_mol.npz, make_golden.py::g_chk_mol): normalised shell tables, AO values, ECP
channel functions, coordinates, charges, lattice.  The image has no HDF5 library, so the byte-scan path is what runs here; the
h5py path is covered where h5py exists."""

import json
import os

import numpy as np
import pytest

from helpers import golden, relerr
from oracle import energy as oenergy
from oracle import gto
from pyqmc_amd import chkfile, tables

FILES = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "files")


@pytest.mark.parametrize("name", ["diamond_primitive", "li_cubic_ccecp"])
def test_chkfile_mol_gives_the_reference_tables(name):
    g = golden("g28_chk_mol")
    mol = chkfile.load_mol(os.path.join(FILES, name + ".hdf5"), backend="scan")
    assert [mol.atom_pure_symbol(i) for i in range(mol.natm)] == [str(s) for s in g[name + "_syms"]]
    assert np.array_equal(mol.atom_coords(), g[name + "_xyz"]) and np.array_equal(mol.atom_charges(), g[name + "_charges"])
    assert relerr(mol.lattice_vectors(), g[name + "_lattice_bohr"]) < 1e-15
    assert sum(mol.nelec) == int(g[name + "_charges"].sum()) and mol.nelec[0] == mol.nelec[1]
    # shell tables as the device gets them (tables.basis_tables) == the reference's AtomicOrbitalEvaluator tables
    t = tables.basis_tables(mol)
    assert np.array_equal(t["shell_l"], g[name + "_basis_ls"]) and np.array_equal(t["shell_prim_off"], g[name + "_splits"])
    assert np.array_equal(t["prim_exp"], g[name + "_basis_arrays"][:, 0])
    assert relerr(t["prim_coef"], g[name + "_basis_arrays"][:, 1]) < 1e-14
    # ... and they evaluate to the reference's AO values (oracle evaluator on the ingested molecule)
    tab = gto.AOTable(mol)
    ao = gto.eval_ao(tab, g[name + "_pts"], 5)
    assert relerr(ao[0], g[name + "_ao"]) < 1e-12
    ref_lap = g[name + "_ao_lap"]  # (5, npts, nao): value, gradient, Laplacian rows of GTOval_sph_deriv2
    assert relerr(ao, ref_lap) < 1e-12
    # ECP channels: the oracle's v_l from the ingested tables == the reference's functors
    for sym in mol._ecp:
        ls = [int(x) for x in g[f"{name}_ecp_{sym}_l"]]
        v = oenergy.v_l(oenergy.ecp_channels(mol._ecp[sym]), g[name + "_ecp_r"])  # columns l = 0.. then the local channel last, like the reference
        assert v.shape == g[f"{name}_ecp_{sym}_v"].shape and sorted(ls) == sorted(ch[0] for ch in mol._ecp[sym][1])
        assert relerr(v, g[f"{name}_ecp_{sym}_v"]) < 1e-13


def test_scan_and_h5py_backends_agree_and_errors_are_loud(tmp_path):
    path = os.path.join(FILES, "diamond_primitive.hdf5")
    d = chkfile.read_mol_json(path, backend="scan")
    assert d["_basis"]["C"][0][0] == 0 and len(d["_atom"]) == 2 and d["exp_to_discard"] == 0.3
    if chkfile.h5py is None:
        with pytest.raises(RuntimeError, match="h5py"):
            chkfile.read_mol_json(path, backend="h5py")
        with pytest.raises(RuntimeError, match="h5py"):
            chkfile.load_scf(path, backend="h5py")
        assert chkfile.load_scf(path)[1].kpts.shape == (8, 3)  # default without h5py: the built-in parser
    else:
        assert chkfile.read_mol_json(path, backend="h5py") == d
        mol, mf = chkfile.load_scf(path)
        assert len(mf.mo_coeff) > 0
        # the built-in parser (hdf5lite) against the real HDF5 library on both of the reference's files: every SCF dataset equal
        for name in ("diamond_primitive", "li_cubic_ccecp"):
            p2 = os.path.join(FILES, name + ".hdf5")
            (m1, f1), (m2, f2) = chkfile.load_scf(p2, backend="h5py"), chkfile.load_scf(p2, backend="lite")
            assert np.array_equal(f1.kpts, f2.kpts) and f1.e_tot == f2.e_tot and m1.nelec == m2.nelec
            c1, c2 = np.asarray(f1.mo_coeff), np.asarray(f2.mo_coeff)
            o1, o2 = np.asarray(f1.mo_occ), np.asarray(f2.mo_occ)
            assert c1.shape == c2.shape and c1.dtype == c2.dtype and np.array_equal(c1, c2) and np.array_equal(o1, o2)
    junk = tmp_path / "junk.bin"
    junk.write_bytes(b"\x89HDF" + b'{"atom": broken' + bytes(100))
    with pytest.raises(ValueError, match="no PySCF mol JSON"):
        chkfile.read_mol_json(str(junk), backend="scan")


@pytest.mark.parametrize("name", ["diamond_primitive", "li_cubic_ccecp"])
def test_hdf5lite_reads_the_reference_checkpoints(name):
    """pyqmc_amd.hdf5lite (no HDF5 library in the images) on the reference's own PySCF checkpoint files: the group tree, the
    variable-length ``mol`` string out of the global heap (equal to what the byte scan finds), real, complex ({r, i} compound)
    and scalar datasets with the shapes a 2x2x2 k-point SCF has."""
    from pyqmc_amd import hdf5lite

    path = os.path.join(FILES, name + ".hdf5")
    f = hdf5lite.File(path)
    assert f.keys() == ["mol", "scf"] and f.is_group("scf") and not f.is_group("mol")
    assert f.keys("scf") == ["e_tot", "kpts", "mo_coeff__from_list__", "mo_energy__from_list__", "mo_occ__from_list__"]
    assert json.loads(f["mol"]) == chkfile.read_mol_json(path, "scan") == chkfile.read_mol_json(path, "lite")
    assert f["scf/kpts"].shape == (8, 3) and f["scf/kpts"].dtype == np.float64 and np.all(f["scf/kpts"][0] == 0.0)
    assert isinstance(float(f["scf/e_tot"]), float) and f["scf/e_tot"] < 0
    nao = {"diamond_primitive": 18, "li_cubic_ccecp": 20}[name]
    assert f.keys("scf/mo_coeff__from_list__") == [f"{k:06d}" for k in range(8)]
    for k in range(8):
        c, o, e = f[f"scf/mo_coeff__from_list__/{k:06d}"], f[f"scf/mo_occ__from_list__/{k:06d}"], f[f"scf/mo_energy__from_list__/{k:06d}"]
        assert c.shape == (nao, nao) and c.dtype == np.complex128 and o.shape == e.shape == (nao,)
        assert np.all(np.diff(e) > -1e-9) and set(np.unique(o)) <= {0.0, 1.0, 2.0}
    assert "scf/mo_coeff" not in f and "scf/e_tot" in f and len(f.walk("/")) == 3 + 3 * 8
    with pytest.raises(KeyError):
        f["scf/nothing"]


def test_load_scf_gives_orbitals_orthonormal_in_our_ao_metric():
    """``chkfile.load_scf`` on the diamond checkpoint: a ``KMeanField`` in the layout ``generate_wf`` takes (spin-duplicated
    KRKS orbitals, occupations occ > 0 / occ > 1 as ``mf.to_uhf()`` gives, pyscftools.py:139-146) whose k-points are exactly the
    Gamma-folding k-points of the 2x2x2 supercell — and, the point of the exercise, orbitals that are ORTHONORMAL under the
    overlap of the oracle's lattice-summed AOs, C_k^H S_k C_k = 1: PySCF's AO convention (normalisation, m-ordering, Bloch
    phase) is the one the reference's in-repo evaluator and this code follow.  (S_k by midpoint quadrature over the cell.)"""
    from oracle import pbc as opbc
    from pyqmc_amd import pbc

    cell, mf = chkfile.load_scf(os.path.join(FILES, "diamond_primitive.hdf5"), backend="lite")
    assert type(mf).__name__ == "KMeanField" and cell.nelec == (4, 4) and abs(mf.e_tot + 10.507476186847358) < 1e-12
    assert all(np.array_equal(mf.mo_occ[s][k], np.r_[np.ones(4), np.zeros(14)]) for s in (0, 1) for k in range(8))
    sup = pbc.get_supercell(cell, 2.0 * np.eye(3))
    want = pbc.get_supercell_kpts(sup)
    frac = (mf.kpts[:, None, :] - want[None, :, :]) @ np.linalg.inv(cell.reciprocal_vectors())
    assert np.all(np.min(np.abs(frac - np.round(frac)).max(axis=2), axis=1) < 1e-8)  # every file k-point folds onto Gamma of the supercell
    lat = cell.lattice_vectors()
    n = 18
    g = (np.stack(np.meshgrid(*[np.arange(n)] * 3, indexing="ij"), -1).reshape(-1, 3) + 0.5) / n
    pt = opbc.PeriodicAOTable(cell, mf.kpts, pbc.lattice_points_within(lat, 30.0), precision=1e-8)
    ao = opbc.eval_ao_pbc(pt, g @ lat, 1)[:, 0]
    w = abs(np.linalg.det(lat)) / len(g)
    for k in range(8):
        C = mf.mo_coeff[0][k]
        assert np.abs(C.conj().T @ (w * ao[k].conj().T @ ao[k]) @ C - np.eye(18)).max() < 1e-5, k


def test_chkfile_reader_handles_labelled_atoms_kappa_and_stale_json():
    """`_basis` / `_ecp` keyed by the atom label as given; a kappa entry refused with the intended message; a file
    holding two different mol strings refused by the byte scan."""
    d = chkfile.read_mol_json(os.path.join(FILES, "diamond_primitive.hdf5"), "lite")
    lab = json.loads(json.dumps(d))
    lab["_atom"] = [["C1", lab["_atom"][0][1]], ["C", lab["_atom"][1][1]]]
    lab["_basis"] = {"C1": d["_basis"]["C"][:2], "C": d["_basis"]["C"]}
    m = chkfile.mol_from_json(lab)
    assert [m.atom_symbol(i) for i in range(2)] == ["C1", "C"] and len(m._basis["C1"]) == 2 and len(m._basis["C"]) == 3
    kap = json.loads(json.dumps(d))
    kap["_basis"]["C"][0] = [0, -1] + kap["_basis"]["C"][0][1:]
    with pytest.raises(NotImplementedError, match="kappa"):
        chkfile.mol_from_json(kap)
    # round 6: a generally contracted shell (two coefficient columns) and a kappa = 0 entry are ingested — the shell becomes two
    # single-column shells in PySCF's AO order, zero coefficients dropped (tables.split_general_contractions)
    gen = json.loads(json.dumps(d))
    first = gen["_basis"]["C"][0]
    gen["_basis"]["C"][0] = [first[0], 0] + [[p[0], p[1], (1.0 if i == len(first) - 2 else 0.0)] for i, p in enumerate(first[1:])]
    mg = chkfile.mol_from_json(gen)
    m0 = chkfile.mol_from_json(d)
    assert len(mg._basis["C"]) == len(m0._basis["C"]) + 1 and mg._basis["C"][0] == m0._basis["C"][0]
    assert mg._basis["C"][1] == [first[0], [float(first[-1][0]), 1.0]] and mg.nao() == m0.nao() + m0.natm * (2 * first[0] + 1)
    raw = open(os.path.join(FILES, "diamond_primitive.hdf5"), "rb").read()
    other = json.dumps({"atom": "x", "_atom": []}).encode()
    assert chkfile._scan_json(raw + b"\0" + raw) == d
    with pytest.raises(ValueError, match="different mol JSON"):
        chkfile._scan_json(other + b"\0" * 8 + raw)


@pytest.mark.parametrize("name,nelec,nao", [("h_pbc_casscf", (1, 1), 10), ("h_noncubic_sto3g_triplet", (2, 0), 2)])
def test_load_scf_single_k_and_unrestricted_layouts(name, nelec, nao):
    """The reference's two other checkpoint fixtures: a single-k-point restricted SCF of a cell (``scf/kpt`` + plain ``mo_coeff``)
    and an unrestricted k-point SCF (``mo_coeff__from_list__/00000S__from_list__/00000K``, here a spin triplet with NO down
    electron) — both come back as one-k-point ``KMeanField``s whose orbitals are orthonormal in the oracle's AO metric."""
    from oracle import pbc as opbc
    from pyqmc_amd import pbc

    cell, mf = chkfile.load_scf(os.path.join(FILES, name + ".hdf5"), backend="lite")
    assert type(mf).__name__ == "KMeanField" and cell.nelec == nelec and mf.kpts.shape == (1, 3) and np.all(mf.kpts == 0.0)
    assert [m[0].shape for m in mf.mo_coeff] == [(nao, nao)] * 2 and [int(o[0].sum()) for o in mf.mo_occ] == list(nelec)
    lat = cell.lattice_vectors()
    n = 40
    g = (np.stack(np.meshgrid(*[np.arange(n)] * 3, indexing="ij"), -1).reshape(-1, 3) + 0.5) / n
    pt = opbc.PeriodicAOTable(cell, mf.kpts, pbc.lattice_points_within(lat, 40.0), precision=1e-8)
    ao = opbc.eval_ao_pbc(pt, g @ lat, 1)[0, 0]
    S = abs(np.linalg.det(lat)) / len(g) * ao.T @ ao
    for s in (0, 1):
        C = mf.mo_coeff[s][0]
        assert np.abs(C.conj().T @ S @ C - np.eye(nao)).max() < 5e-6, s  # (quadrature of the 13.01-exponent s function limits this)


@pytest.mark.parametrize("unrestricted", [False, True])
def test_load_scf_k_point_orbitals_stored_as_arrays(unrestricted, monkeypatch):
    """A k-point SCF whose orbitals PySCF stored as arrays (``scf/mo_coeff`` (nk, nao, nmo) / ``scf/mo_occ`` (nk, nmo); unrestricted
    (2, nk, ...)) next to ``scf/kpts`` — equal orbital counts at every k — must come back as the same [spin][k] lists as the
    ``__from_list__`` layout of the reference's diamond checkpoint, never as the single-k-point branch with k = 0 and k = 1 taken for
    the two spin channels.  Shapes that fit neither layout raise."""
    path = os.path.join(FILES, "diamond_primitive.hdf5")
    cell, ref = chkfile.load_scf(path, backend="lite")
    nk = len(ref.kpts)
    mo = np.stack([ref.mo_coeff[0][k] for k in range(nk)])
    occ2 = np.stack([ref.mo_occ[0][k] + ref.mo_occ[1][k] for k in range(nk)])  # restricted occupations 0 / 2
    data = {"scf/kpts": np.asarray(ref.kpts), "scf/e_tot": np.float64(-1.0)}
    if unrestricted:
        data["scf/mo_coeff"] = np.stack([mo, mo])
        data["scf/mo_occ"] = np.stack([np.stack(ref.mo_occ[0]), np.stack(ref.mo_occ[1])])
    else:
        data["scf/mo_coeff"], data["scf/mo_occ"] = mo, occ2
    monkeypatch.setattr(chkfile, "load_mol", lambda p, b=None: cell)
    monkeypatch.setattr(chkfile, "_open", lambda p, b=None: (data, lambda n: data[n], lambda n: [], lambda: None))
    _, mf = chkfile.load_scf("unused", backend="lite")
    assert type(mf).__name__ == "KMeanField" and np.array_equal(mf.kpts, ref.kpts)
    for s in (0, 1):
        assert len(mf.mo_coeff[s]) == nk
        for k in range(nk):
            assert np.array_equal(mf.mo_coeff[s][k], ref.mo_coeff[s][k]) and np.array_equal(mf.mo_occ[s][k], ref.mo_occ[s][k])
    data["scf/mo_occ"] = np.asarray(data["scf/mo_occ"]).reshape(-1)  # neither layout
    with pytest.raises(NotImplementedError):
        chkfile.load_scf("unused", backend="lite")
