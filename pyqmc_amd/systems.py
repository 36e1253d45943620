"""Plain-array descriptions of trial-wave-function inputs.

The reference takes PySCF ``Mole``/``SCF`` objects (``pyqmc/pyscftools.py:105-191``).
PySCF is not available where this package runs, so the hot path is fed from the
small duck-typed containers below.  They expose exactly the attributes the
reference's hot path reads from a PySCF molecule (``_basis``, ``_ecp``, ``_atom``,
``nelec``, ``natm``, ``cart``, ``atom_coords()``, ``atom_charges()``,
``atom_pure_symbol()``, ``atom_symbol()``, ``has_ecp()``; see
``pyqmc/wf/numba/gto.py:435-470``, ``pyqmc/observables/eval_ecp.py:21-40``,
``pyqmc/wf/jastrowspin.py:31-54``, ``pyqmc/method/mc.py:25-73``) so the same
object can be handed to the reference itself when golden vectors are generated.

All basis/ECP tables here are synthetic but ccECP/cc-pVDZ shaped (same shell
structure and contraction lengths); there is no SCF available, so MO
coefficients are seeded random orthonormal columns.  Wave-function quality is
irrelevant to throughput, and parity always uses identical inputs on both
sides.
"""

import numpy as np

# --- synthetic cc-pVDZ-shaped tables (pyscf ``_basis`` format: [l, [exp, coef], ...]) ---
_O_BASIS = [
    [0, [54.775216, -0.0012444], [25.616801, 0.0107330], [11.980245, 0.0018889],
        [6.992317, -0.1742537], [2.620277, 0.0017622], [1.225429, 0.3161846],
        [0.577797, 0.4512023], [0.268022, 0.3121534], [0.125346, 0.0511167]],
    [0, [0.258551, 1.0]],
    [1, [22.217266, 0.0104866], [10.747550, 0.0366435], [5.315785, 0.0803674],
        [2.660761, 0.1627010], [1.331816, 0.2377791], [0.678626, 0.2811422],
        [0.333673, 0.2643189], [0.167017, 0.1466014], [0.083598, 0.0458145]],
    [1, [0.267865, 1.0]],
    [2, [1.232753, 1.0]],
]
_H_BASIS = [
    [0, [23.843185, 0.0041149], [10.212443, 0.0104644], [4.374164, 0.0280111],
        [1.873529, 0.0758862], [0.802465, 0.1821062], [0.343709, 0.3485214],
        [0.147217, 0.3782313], [0.063055, 0.1164241]],
    [0, [0.139013, 1.0]],
    [1, [0.740212, 1.0]],
]
_HE_BASIS = [
    [0, [39.320931, 0.0100657], [17.174528, 0.0248762], [7.501461, 0.0582537],
        [3.276475, 0.1345596], [1.431093, 0.2443118], [0.625070, 0.3425061],
        [0.273017, 0.2897371], [0.119248, 0.0685519]],
    [0, [0.394193, 1.0]],
    [1, [1.347921, 1.0]],
]
_C_BASIS = [
    [0, [13.073594, 0.0051583], [6.541187, 0.0603424], [4.573411, -0.1978471],
        [1.637494, -0.081034], [0.819297, 0.2321726], [0.409924, 0.2914643]],
    [0, [0.127852, 1.0]],
    [1, [9.934169, 0.0209076], [3.886955, 0.0572698], [1.871016, 0.1122682],
        [0.935757, 0.2130082], [0.468003, 0.2835815]],
    [1, [0.149161, 1.0]],
    [2, [0.56116, 1.0]],
]
# ECP in pyscf ``_ecp`` format: {sym: (ncore, [[l, [terms r^-2, r^-1, r^0, r^1, ...]], ...])}
# l = -1 is the local channel.  (eval_ecp.py:160-179 reads index n as r^(n-2).)
_ECP = {
    "O": (2, [[-1, [[], [[12.30997, 6.0]], [[13.71419, -47.876]], [[14.76962, 73.85984]]]],
              [0, [[], [], [[13.65512, 85.86406]]]]]),
    "H": (0, [[-1, [[], [[21.24359508, 1.0]], [[21.77696655, -10.85192405]], [[21.24359508, 21.24359508]]]],
              [0, [[], [], [[1.0, 0.0]]]]]),
    "He": (0, [[-1, [[], [[32.0, 2.0]], [[33.713355, -27.700840]], [[32.0, 64.0]]]],
               [0, [[], [], [[1.0, 0.0]]]]]),
    "C": (2, [[-1, [[], [[14.43502, 4.0]], [[7.38188, -25.81955]], [[8.39889, 57.74008]]]],
              [0, [[], [], [[7.76079, 52.13345]]]]]),
}
_BASIS = {"O": _O_BASIS, "H": _H_BASIS, "He": _HE_BASIS, "C": _C_BASIS}
_VALENCE = {"O": 6, "H": 1, "He": 2, "C": 4}


class Mol:
    """Duck-typed stand-in for the subset of ``pyscf.gto.Mole`` the hot path reads."""

    cart = False

    def __init__(self, symbols, coords_bohr, nelec=None, basis=None, ecp=None, charges=None):
        self._names = list(symbols)
        self._coords = np.asarray(coords_bohr, dtype=float).reshape(-1, 3)
        self._basis = {k: (basis or _BASIS)[k] for k in dict.fromkeys(self._names)}
        src_ecp = _ECP if ecp is None else ecp
        self._ecp = {k: src_ecp[k] for k in dict.fromkeys(self._names) if k in src_ecp}
        self._atom = [(n, [float(x) for x in c]) for n, c in zip(self._names, self._coords)]
        self.natm = len(self._names)
        if charges is None:
            charges = [_VALENCE[n] for n in self._names]
        self._q = np.asarray(charges, dtype=float)
        if nelec is None:
            ne = int(round(self._q.sum()))
            nelec = (ne - ne // 2, ne // 2)
        self.nelec = (int(nelec[0]), int(nelec[1]))

    def atom_coords(self):
        return self._coords

    def atom_charges(self):
        return self._q

    def atom_pure_symbol(self, i):
        return self._names[i]

    def atom_symbol(self, i):
        return self._names[i]

    def has_ecp(self):
        return bool(self._ecp)

    # -- derived sizes ---------------------------------------------------
    def nao(self):
        return int(sum(sum(2 * sh[0] + 1 for sh in self._basis[n]) for n in self._names))


class Cell(Mol):
    """Duck-typed stand-in for the subset of ``pyscf.pbc.gto.Cell`` the hot path reads: ``Mol`` plus the
    lattice (``a``, ``lattice_vectors()``, ``reciprocal_vectors()``, ``vol``; rows are lattice vectors in
    bohr).  ``hasattr(cell, "a")`` is what switches the reference to ``PeriodicConfigs``/Ewald
    (``mc.py:69``, ``accumulators.py:52``, ``wftools.py:82-83``)."""

    def __init__(self, symbols, coords_bohr, a, **kw):
        super().__init__(symbols, coords_bohr, **kw)
        self.a = np.asarray(a, dtype=float).reshape(3, 3)
        self.dimension = 3

    def lattice_vectors(self):
        return self.a

    def reciprocal_vectors(self):
        return 2 * np.pi * np.linalg.inv(self.a).T

    @property
    def vol(self):
        return float(abs(np.linalg.det(self.a)))


class MeanField:
    """Duck-typed UHF-like container (``mo_coeff (2,nao,nmo)``, ``mo_occ (2,nmo)``)."""

    def __init__(self, mo_coeff, mo_occ):
        self.mo_coeff = np.asarray(mo_coeff)
        self.mo_occ = np.asarray(mo_occ)

    def to_uhf(self, *a):
        return self


# water monomer, bohr (r_OH = 1.8088, angle 104.5 deg)
_WATER = [("O", (0.0, 0.0, 0.0)), ("H", (0.0, -1.4304, 1.1072)), ("H", (0.0, 1.4304, 1.1072))]


def water():
    """C2 of BASELINE.json: H2O, 8 valence electrons, 23 AOs."""
    sym, xyz = zip(*_WATER)
    return Mol(sym, xyz)


def water_cluster(nx=2, ny=2, nz=2, spacing=6.0):
    """(H2O)_n on a grid; 2x2x2 is the 64-electron metric system (24 atoms, 184 AOs)."""
    sym, xyz = [], []
    for ix in range(nx):
        for iy in range(ny):
            for iz in range(nz):
                shift = np.array([ix, iy, iz], dtype=float) * spacing
                for s, c in _WATER:
                    sym.append(s)
                    xyz.append(np.asarray(c) + shift)
    return Mol(sym, xyz)


def helium():
    """C1 of BASELINE.json: He atom, 2 electrons, 5 AOs."""
    return Mol(["He"], [(0.0, 0.0, 0.0)])


def carbon_dimer(r=2.35):
    return Mol(["C", "C"], [(0.0, 0.0, 0.0), (0.0, 0.0, r)])


def carbon_dimer_high_l(r=2.35):
    """C2 with f, g and h shells added to the carbon tables (one contracted g shell): exercises l = 3..5, the range of
    the reference's evaluator (numba/gto.py:107-118) beyond the double-zeta shells of the BASELINE configurations."""
    extra = [[3, [0.761, 1.0]], [4, [2.1, 0.6], [0.7, 0.5]], [4, [0.43, 1.0]], [5, [1.05, 1.0]]]
    return Mol(["C", "C"], [(0.0, 0.0, 0.0), (0.3, -0.2, r)], basis={"C": _C_BASIS + extra})


_DIAMOND_A = 3.5668 / 0.529177210903  # conventional cubic lattice constant, bohr (benchmarks/c_solid_benchmark.py)
_DIAMOND_FRAC = [(0, 0, 0), (0, .5, .5), (.5, 0, .5), (.5, .5, 0),
                 (.25, .25, .25), (.25, .75, .75), (.75, .25, .75), (.75, .75, .25)]


def diamond_cubic(n=1):
    """Diamond, conventional cubic cell repeated n x n x n (8 n^3 C atoms, 32 n^3 valence electrons) —
    the shape of BASELINE.json's C3 (``benchmarks/c_solid_benchmark.py``).  Diagonal lattice."""
    a = _DIAMOND_A
    xyz = [(np.array(f) + np.array([i, j, k])) * a for i in range(n) for j in range(n) for k in range(n)
           for f in _DIAMOND_FRAC]
    return Cell(["C"] * len(xyz), xyz, np.eye(3) * a * n)


def diamond_primitive():
    """Diamond primitive (fcc) cell: 2 C atoms, 8 valence electrons, non-orthogonal lattice vectors."""
    a = _DIAMOND_A
    lat = 0.5 * a * np.array([[0., 1., 1.], [1., 0., 1.], [1., 1., 0.]])
    return Cell(["C", "C"], [(0, 0, 0), (a / 4, a / 4, a / 4)], lat)


def random_mf(mol, seed=20260928, nvirt=0, scale_virtual=1.0):
    """Seeded MO coefficients: first columns of qr(randn(nao,nao)); occupied = lowest n_s.

    ``nvirt`` extra (unoccupied) columns are kept so multi-determinant expansions
    have somewhere to excite into.
    """
    rng = np.random.default_rng(seed)
    nao = mol.nao()
    nmo = min(nao, max(mol.nelec) + nvirt)
    mo = np.empty((2, nao, nmo))
    occ = np.zeros((2, nmo))
    for s in range(2):
        q, _ = np.linalg.qr(rng.standard_normal((nao, nao)))
        mo[s] = q[:, :nmo]
        occ[s, : mol.nelec[s]] = 1.0
    return MeanField(mo, occ)


def random_determinants(mol, mf, ndet, seed=7):
    """[(coef, [occ_up, occ_dn]), ...] single/double excitations out of the aufbau determinant
    (the list format of ``determinant_tools.create_packed_objects``, determinant_tools.py:39-71)."""
    rng = np.random.default_rng(seed)
    nmo = mf.mo_coeff.shape[-1]
    base = [list(range(mol.nelec[0])), list(range(mol.nelec[1]))]
    dets = [(1.0, [list(base[0]), list(base[1])])]
    seen = {(tuple(base[0]), tuple(base[1]))}
    guard = 0
    while len(dets) < ndet and guard < 100000:
        guard += 1
        occ = [list(base[0]), list(base[1])]
        for s in range(2):
            if mol.nelec[s] == 0 or nmo == mol.nelec[s] or rng.random() < 0.35:
                continue
            i = rng.integers(mol.nelec[s])
            a = rng.integers(mol.nelec[s], nmo)
            occ[s][i] = int(a)
            occ[s] = sorted(occ[s])
        key = (tuple(occ[0]), tuple(occ[1]))
        if key in seen:
            continue
        seen.add(key)
        dets.append((float(0.3 * rng.standard_normal()), occ))
    return dets


def initial_guess(mol, nconfig, r=1.0, rng=None):
    """Electrons near atoms in proportion to charge — semantics of ``mc.initial_guess``
    (``pyqmc/method/mc.py:25-73``), with an explicit generator instead of the global one."""
    from pyqmc_amd.configs import OpenConfigs, PeriodicConfigs

    rng = np.random.default_rng(1234) if rng is None else rng
    epos = np.zeros((nconfig, int(np.sum(mol.nelec)), 3))
    wts = mol.atom_charges() / np.sum(mol.atom_charges())
    for s in (0, 1):
        neach = np.floor(mol.nelec[s] * wts).astype(int)
        nassigned = int(neach.sum())
        totleft = int(mol.nelec[s] - nassigned)
        ind0 = s * mol.nelec[0]
        epos[:, ind0 : ind0 + nassigned, :] = np.repeat(mol.atom_coords(), neach, axis=0)
        if totleft > 0:
            inds = np.argpartition(rng.random((nconfig, len(wts))), totleft, axis=1)[:, :totleft]
            epos[:, ind0 + nassigned : ind0 + mol.nelec[s], :] = mol.atom_coords()[inds]
    epos += r * rng.standard_normal(epos.shape)
    if hasattr(mol, "a"):  # mc.py:69-72
        return PeriodicConfigs(epos, mol.lattice_vectors())
    return OpenConfigs(epos)
