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
    """First balanced JSON object starting with ``start_key`` in ``raw`` (bytes), string-aware brace matching."""
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
                        return json.loads(raw[pos : i + 1].decode("utf-8"))
                    except (UnicodeDecodeError, json.JSONDecodeError):
                        break
            i += 1
        pos = raw.find(start_key, pos + 1)
    raise ValueError("no PySCF mol JSON found in the file")


def read_mol_json(path, backend=None):
    """The ``mol`` JSON of a PySCF chkfile as a dict.  ``backend``: "h5py", "scan" or None (h5py when importable)."""
    backend = backend or ("h5py" if h5py is not None else "scan")
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
    basis = {k: [[int(sh[0])] + [[float(x) for x in p] for p in sh[1:]] for sh in v] for k, v in d["_basis"].items()}
    for sym, shells in basis.items():
        for sh in shells:
            if not all(len(p) == 2 for p in sh[1:]):
                raise NotImplementedError(f"{sym}: general contractions (several coefficient columns, or a kappa entry) are not supported")
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
    kw = dict(nelec=nelec, basis={s: basis[s] for s in dict.fromkeys(pure)}, ecp=ecp, charges=charges)
    if d.get("a") is None:
        m = Mol(pure, coords, **kw)
    else:
        unit = str(d.get("unit", "angstrom") or "angstrom").lower()
        a = np.array(d["a"], dtype=float).reshape(3, 3)
        if not unit.startswith(("b", "au")):  # Cell.lattice_vectors(): a / BOHR unless the input unit was bohr
            a = a / PYSCF_BOHR
        m = Cell(pure, coords, a, **kw)
    m.exp_to_discard = d.get("exp_to_discard")
    m.precision = d.get("precision")
    m.basis_name, m.ecp_name = d.get("basis"), d.get("ecp")
    return m


def load_mol(path, backend=None):
    """Molecule or cell of a PySCF chkfile (``pyscf.lib.chkfile.load_mol`` / ``load_cell`` for the attributes the hot path reads)."""
    return mol_from_json(read_mol_json(path, backend))


def load_scf(path):
    """``(mol, MeanField)`` with the chkfile's ``scf/mo_coeff`` and ``scf/mo_occ`` (needs h5py: binary datasets).  Restricted
    results are duplicated to the two spin channels like ``mf.to_uhf()`` (pyscftools.py:139-146); k-point lists stay lists."""
    if h5py is None:
        raise RuntimeError("reading MO coefficients from a chkfile needs h5py (binary HDF5 datasets); the mol JSON does not")
    from .systems import MeanField

    mol = load_mol(path, "h5py")
    with h5py.File(path, "r") as f:
        g = f["scf"]
        if "mo_coeff" in g:
            mo, occ = np.array(g["mo_coeff"]), np.array(g["mo_occ"])
            if mo.ndim == 2:  # RHF / ROHF: (nao, nmo), occupations 0 / 1 / 2
                mo = np.stack([mo, mo])
                occ = np.stack([(occ > 0).astype(float), (occ > 1).astype(float)])
            mf = MeanField(mo, occ)
        else:  # k-point SCF: lists stored as mo_coeff__from_list__/000000 ...
            key = lambda name: [np.array(g[name][k]) for k in sorted(g[name])]
            mf = MeanField.__new__(MeanField)
            mf.mo_coeff, mf.mo_occ = key("mo_coeff__from_list__"), key("mo_occ__from_list__")
            mf.kpts = np.array(g["kpts"]) if "kpts" in g else None
    return mol, mf
