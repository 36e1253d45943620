"""Read the molecule / cell description out of a PySCF checkpoint file — SURVEY.md section 8(f4), second half.

The reference starts every calculation from a chkfile (``pyqmc/pyscftools.py:105-191`` ``recover_pyscf`` ->
``pyscf.lib.chkfile.load_mol`` / ``load_cell``): the dataset ``mol`` of that HDF5 file is ONE JSON string, PySCF's
``Mole.dumps()``, carrying everything the hot path reads from a molecule — ``_atom`` (symbols, coordinates in bohr), ``_basis``
(``{symbol: [[l, [exp, coef], ...], ...]}`` AFTER ``exp_to_discard``), ``_ecp`` (``{symbol: [ncore, [[l, [terms r^-2 .. r^4]], ..]]}``),
``_atm`` (effective nuclear charges), ``spin`` / ``charge`` and, for cells, ``a`` (lattice vectors in the INPUT unit).
``load_mol`` turns it into the duck-typed ``pyqmc_amd.systems.Mol`` / ``Cell`` that ``tables.py``, ``generate_wf`` and the
accumulators take, so a real SCF output can feed the device path instead of the synthetic tables.

Two ways to get at the string: with ``h5py`` importable, ``f["mol"][()]``; without it (this image has no HDF5 library) the file's
bytes are scanned for the JSON object — HDF5 stores a scalar string dataset contiguously, and the object is recognised by its
leading ``{"atom":`` key and balanced braces.  The MO coefficients (``scf/mo_coeff`` ...) are binary datasets and need h5py.
"""

import json

import numpy as np

try:  # optional: not in this image
    import h5py
except ImportError:
    h5py = None

PYSCF_BOHR = 0.52917721092  # pyscf.data.nist.BOHR (angstrom): the constant PySCF converted the chkfile's coordinates with


def _scan_json(raw, start_key=b'{"atom"'):
    """The balanced JSON object starting with ``start_key`` in ``raw`` (bytes), string-aware brace matching.  A chkfile an SCF was
    re-run into can still hold an older ``mol`` string in freed space: several DIFFERENT objects are an error (use h5py or
    ``hdf5lite``, which follow the file's own pointers); identical copies are one."""
    found = []
    pos = raw.find(start_key)
    while pos >= 0:
        depth, i, in_str, esc = 0, pos, False, False
        while i < len(raw):
            c = raw[i]
            if in_str:
                if esc:
                    esc = False
                elif c == 0x5C:  # backslash
                    esc = True
                elif c == 0x22:
                    in_str = False
            elif c == 0x22:
                in_str = True
            elif c == 0x7B:
                depth += 1
            elif c == 0x7D:
                depth -= 1
                if depth == 0:
                    try:
                        found.append(json.loads(raw[pos : i + 1].decode("utf-8")))
                    except (UnicodeDecodeError, json.JSONDecodeError):
                        pass
                    break
            i += 1
        pos = raw.find(start_key, pos + 1)
    if not found:
        raise ValueError("no PySCF mol JSON found in the file")
    if any(o != found[-1] for o in found):
        raise ValueError(f"{len(found)} different mol JSON objects in the file (a rewritten chkfile): read it with backend='lite' or h5py")
    return found[-1]


def read_mol_json(path, backend=None):
    """The ``mol`` JSON of a PySCF chkfile as a dict.  ``backend``: "h5py", "lite" (``pyqmc_amd.hdf5lite``: follows the file's
    group and heap pointers), "scan" (byte scan for the object) or None (h5py when importable, else lite)."""
    backend = backend or ("h5py" if h5py is not None else "lite")
    if backend == "lite":
        from .hdf5lite import File

        return json.loads(File(path)["mol"])
    if backend == "h5py":
        if h5py is None:
            raise RuntimeError("h5py is not installed: use backend='scan'")
        with h5py.File(path, "r") as f:
            s = f["mol"][()]
        return json.loads(s.decode() if isinstance(s, bytes) else s)
    with open(path, "rb") as f:
        return _scan_json(f.read())


def mol_from_json(d):
    """``systems.Mol`` (no lattice) or ``systems.Cell`` from the dict of ``read_mol_json``."""
    from .systems import Cell, Mol

    symbols = [a[0] for a in d["_atom"]]
    coords = np.array([a[1] for a in d["_atom"]], dtype=float)  # bohr (Mole.build converts; format_atom, pyscf/gto/mole.py)
    pure = ["".join(ch for ch in s if ch.isalpha()) for s in symbols]  # atom_pure_symbol: 'C1' -> 'C'
    # (generally contracted shells and kappa = 0 entries are taken apart by systems.Mol: tables.split_general_contractions)
    basis = {k: [[int(sh[0])] + [(int(p) if not hasattr(p, "__len__") else [float(x) for x in p]) for p in sh[1:]] for sh in v] for k, v in d["_basis"].items()}
    ecp = {k: (int(v[0]), [[int(ch[0]), [[[float(t[0]), float(t[1])] for t in terms] for terms in ch[1]]] for ch in v[1]]) for k, v in (d.get("_ecp") or {}).items()}
    # effective nuclear charges: _atm[:, 0] (CHARGE_OF) already has the ECP core removed
    if d.get("_atm"):
        charges = [float(row[0]) for row in d["_atm"]]
    else:
        raise ValueError("mol JSON without _atm: cannot determine the nuclear charges")
    ntot = int(round(sum(charges))) - int(d.get("charge", 0) or 0)
    spin = int(d.get("spin", 0) or 0)
    if (ntot + spin) % 2:
        raise ValueError(f"{ntot} electrons cannot have spin {spin}")
    nelec = ((ntot + spin) // 2, (ntot - spin) // 2)
    # PySCF keys _basis / _ecp by the atom label as given ('C1', 'ghost-H') and falls back to the pure symbol
    by_label = lambda table, lab, p: table[lab] if lab in table else table[p]
    labels = list(dict.fromkeys(zip(symbols, pure)))
    if any(lab != p and (lab in basis or lab in ecp) for lab, p in labels):  # labelled species: each label is its own kind
        names = symbols
        kw = dict(nelec=nelec, basis={lab: by_label(basis, lab, p) for lab, p in labels}, ecp={lab: by_label(ecp, lab, p) for lab, p in labels if lab in ecp or p in ecp}, charges=charges)
    else:
        names = pure
        kw = dict(nelec=nelec, basis={s: basis[s] for s in dict.fromkeys(pure)}, ecp=ecp, charges=charges)
    if d.get("a") is None:
        m = Mol(names, coords, **kw)
    else:
        unit = str(d.get("unit", "angstrom") or "angstrom").lower()
        a = np.array(d["a"], dtype=float).reshape(3, 3)
        if not unit.startswith(("b", "au")):  # Cell.lattice_vectors(): a / BOHR unless the input unit was bohr
            a = a / PYSCF_BOHR
        m = Cell(names, coords, a, **kw)
    m.exp_to_discard = d.get("exp_to_discard")
    m.precision = d.get("precision")
    m.basis_name, m.ecp_name = d.get("basis"), d.get("ecp")
    return m


def load_mol(path, backend=None):
    """Molecule or cell of a PySCF chkfile (``pyscf.lib.chkfile.load_mol`` / ``load_cell`` for the attributes the hot path reads)."""
    return mol_from_json(read_mol_json(path, backend))


def _open(path, backend):
    """(file object with ``f[path]``, ``path in f`` and ``keys``, close function) through h5py or the built-in parser."""
    backend = backend or ("h5py" if h5py is not None else "lite")
    if backend == "h5py":
        if h5py is None:
            raise RuntimeError("h5py is not installed: use backend='lite'")
        f = h5py.File(path, "r")
        return f, (lambda name: np.array(f[name])), (lambda name: sorted(f[name])), f.close
    from .hdf5lite import File

    f = File(path)
    return f, (lambda name: np.asarray(f[name])), f.keys, (lambda: None)


def load_scf(path, backend=None):
    """``(mol, mean field)`` with the chkfile's ``scf/mo_coeff`` / ``scf/mo_occ`` (``pyscftools.recover_pyscf`` +
    ``orbital_evaluator_from_pyscf``, pyscftools.py:105-191).  Restricted results are duplicated to the two spin channels like
    ``mf.to_uhf()`` (:139-146: occupations ``occ > 0`` and ``occ > 1``); a k-point SCF (``mo_coeff__from_list__/000000`` ...)
    becomes a ``pbc.KMeanField`` with ``kpts`` — what ``generate_wf(get_supercell(cell, S), mf)`` takes.
    ``backend``: "h5py", "lite" (``pyqmc_amd.hdf5lite``, no HDF5 library needed) or None = h5py when importable."""
    from .pbc import KMeanField
    from .systems import MeanField

    mol = load_mol(path, backend)
    f, arr, keys, close = _open(path, backend)
    try:
        def uhf(mo, occ):
            mo, occ = np.asarray(mo), np.asarray(occ)
            if occ.ndim == 1:  # RHF / ROHF: one set of orbitals, occupations 0 / 1 / 2
                return [mo, mo], [(occ > 0).astype(float), (occ > 1).astype(float)]
            return [mo[0], mo[1]], [occ[0], occ[1]]

        periodic = hasattr(mol, "a") or hasattr(mol, "lattice_vectors")
        if "scf/mo_coeff" in f and periodic and "scf/kpts" in f:
            # a k-point SCF whose orbitals were stored as ARRAYS (equal orbital counts at every k): KRHF mo_coeff (nk, nao, nmo),
            # mo_occ (nk, nmo); KUHF (2, nk, nao, nmo) / (2, nk, nmo) — split along k into the [spin][k] lists, as pyscf's
            # mf.mo_coeff[k] indexing does for lists and arrays alike.  Never read as one k-point.
            mo_a, occ_a, kpts = np.asarray(arr("scf/mo_coeff")), np.asarray(arr("scf/mo_occ")), np.asarray(arr("scf/kpts"), dtype=float).reshape(-1, 3)
            nk = len(kpts)
            if occ_a.ndim == 2 and mo_a.ndim == 3 and len(mo_a) == nk and len(occ_a) == nk:
                per_k = [uhf(mo_a[k], occ_a[k]) for k in range(nk)]
                mf = KMeanField(kpts, [[pk[0][sp] for pk in per_k] for sp in (0, 1)], [[pk[1][sp] for pk in per_k] for sp in (0, 1)])
            elif occ_a.ndim == 3 and mo_a.ndim == 4 and mo_a.shape[:2] == (2, nk) and occ_a.shape[:2] == (2, nk):
                mf = KMeanField(kpts, [[mo_a[sp, k] for k in range(nk)] for sp in (0, 1)], [[occ_a[sp, k] for k in range(nk)] for sp in (0, 1)])
            else:
                raise NotImplementedError(f"k-point SCF with mo_coeff {mo_a.shape} / mo_occ {occ_a.shape} for {nk} k-points")
            mf.mo_energy = None
        elif "scf/mo_coeff" in f:
            mo, occ = uhf(arr("scf/mo_coeff"), arr("scf/mo_occ"))
            if periodic:  # a single-k-point SCF of a cell (pbc.scf.RHF / UHF with kpt): a one-entry k-point list
                kpt = arr("scf/kpt") if "scf/kpt" in f else np.zeros(3)
                mf = KMeanField(np.asarray(kpt, dtype=float).reshape(1, 3), [[mo[0]], [mo[1]]], [[occ[0]], [occ[1]]])
            else:
                mf = MeanField(np.stack(mo), np.stack(occ))
        else:
            names = keys("scf/mo_coeff__from_list__")
            if names and names[0].endswith("__from_list__"):  # unrestricted k-point SCF: [spin][k] nested lists
                ks = keys(f"scf/mo_coeff__from_list__/{names[0]}")
                get = lambda what: [[arr(f"scf/{what}__from_list__/{sp}/{k}") for k in ks] for sp in names[:2]]
                mf = KMeanField(arr("scf/kpts"), get("mo_coeff"), get("mo_occ"))
                mf.mo_energy = get("mo_energy") if "scf/mo_energy__from_list__" in f else None
            else:
                per_k = [uhf(arr(f"scf/mo_coeff__from_list__/{k}"), arr(f"scf/mo_occ__from_list__/{k}")) for k in names]
                mf = KMeanField(arr("scf/kpts"), [[pk[0][s] for pk in per_k] for s in (0, 1)], [[pk[1][s] for pk in per_k] for s in (0, 1)])
                mf.mo_energy = [arr(f"scf/mo_energy__from_list__/{k}") for k in names] if "scf/mo_energy__from_list__" in f else None
        mf.e_tot = float(arr("scf/e_tot")) if "scf/e_tot" in f else None
    finally:
        close()
    return mol, mf
