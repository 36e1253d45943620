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
        from .tables import split_general_contractions

        # (general contractions — several coefficient columns over one set of exponents — become single-column shells in PySCF's AO order)
        self._basis = {k: split_general_contractions((basis or _BASIS)[k]) for k in dict.fromkeys(self._names)}
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


# generally contracted all-electron tables in cc-pVDZ's shape (two s contractions over eight primitives, the p contraction sharing its
# exponents with an uncontracted function written as a second column with zeros): what every all-electron PySCF checkpoint holds
_O_GENERAL = [
    [0, [11720.0, 0.000710, -0.000160], [1759.0, 0.005470, -0.001263], [400.8, 0.027837, -0.006267], [113.7, 0.104800, -0.025716],
        [37.03, 0.283062, -0.070924], [13.27, 0.448719, -0.165411], [5.025, 0.270952, -0.116955], [1.013, 0.015458, 0.557368]],
    [0, [0.3023, 1.0]],
    [1, [17.70, 0.043018, 0.0], [3.854, 0.228913, 0.0], [1.046, 0.508728, 0.0], [0.2753, 0.460531, 1.0]],
    [2, [1.185, 1.0]],
]
_H_GENERAL = [
    [0, [13.01, 0.019685, 0.0], [1.962, 0.137977, 0.0], [0.4446, 0.478148, 0.0], [0.1220, 0.501240, 1.0]],
    [1, [0.727, 1.0]],
]


def water_general():
    """All-electron H2O (5 + 5 electrons, no ECP) in a generally contracted basis: 24 AOs from 6 PySCF shells = 9 single-column shells."""
    sym, xyz = zip(*_WATER)
    return Mol(sym, xyz, basis={"O": _O_GENERAL, "H": _H_GENERAL}, ecp={}, charges=[8.0, 1.0, 1.0])


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


def water_multichannel(lmax=2):
    """H2O whose oxygen carries s, p and d non-local channels (``lmax=4``: also f and g, the last Legendre function the reference
    tabulates, eval_ecp.py:203-225) (four channels with the local one: the reference's default rule
    for it is the 12-point icosahedral grid, eval_ecp.py:239-240) and whose hydrogens keep the one-channel table (6 points):
    the system of the quadrature-rule fixtures (naip = None, 18, 26, 32, 50)."""
    ecp = dict(_ECP)
    ecp["O"] = (2, [[-1, [[], [[12.30997, 6.0]], [[13.71419, -47.876]], [[14.76962, 73.85984]]]],
                    [0, [[], [], [[13.65512, 85.86406]]]],
                    [1, [[], [], [[9.21, -3.4], [2.87, 1.15]]]],
                    [2, [[], [], [[6.4, -1.9]], [[3.1, 0.45]]]]])
    if lmax >= 3:
        ecp["O"][1].append([3, [[], [], [[4.7, 1.3]]]])
    if lmax >= 4:
        ecp["O"][1].append([4, [[], [], [[3.9, -0.8], [1.7, 0.12]]]])
    sym, xyz = zip(*_WATER)
    return Mol(sym, xyz, ecp=ecp)


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


def diamond_primitive_high_l():
    """The diamond primitive cell with a g and an h shell added to the carbon table: lattice-summed shells beyond f — the range
    of the reference's periodic evaluator (numba/pbcgto.py:52-96 wraps SPH0..SPH5)."""
    a = _DIAMOND_A
    lat = 0.5 * a * np.array([[0., 1., 1.], [1., 0., 1.], [1., 1., 0.]])
    extra = [[4, [1.9, 0.6], [0.8, 0.5]], [5, [1.2, 1.0]]]
    return Cell(["C", "C"], [(0, 0, 0), (a / 4, a / 4, a / 4)], lat, basis={"C": _C_BASIS + extra})


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


def model_mf(mol, screen=None, nrad=70, seed=None):
    """Physically shaped orbitals without an SCF program (none exists where this package runs): eigenvectors of a model
    one-electron Hamiltonian  h = -1/2 lap + sum_A [ -Z*_A / r_A + v_loc,A(r_A) ]  in the molecule's AO basis, with Slater-rule
    screened charges Z* (``screen``: {symbol: Z*}; default O 4.55, C 3.25, H 1.0, He 1.7) standing in for the Hartree
    potential and the local ECP channel v_loc of ``mol._ecp``; the lowest n_s eigenvectors per spin are occupied.  Matrix
    elements by Becke-partitioned atom-centred quadrature (Gauss-Chebyshev radial x 50-point octahedral angular grid); the AO
    values and Laplacians come from the same shell tables the device gets, through a plain NumPy evaluation here.

    Any orbitals are a valid trial function and both sides of a parity test use the same coefficients; what this buys over
    ``random_mf`` is a local energy whose standard deviation is ~1 Ha instead of ~5 Ha, so that an energy comparison "within
    statistical error" (north_star) can actually fail.  Returns ``MeanField`` (mo_coeff (2, nao, nmo), all nao orbitals)."""
    from .tables import basis_tables

    screen = {"O": 4.55, "C": 3.25, "H": 1.0, "He": 1.7, **(screen or {})}
    R = np.asarray(mol.atom_coords(), dtype=float)
    na = mol.natm
    # ---- atom-centred grids, Becke weights
    i_ = np.arange(1, nrad + 1)
    xk = np.cos(i_ * np.pi / (nrad + 1))  # Gauss-Chebyshev (second kind) mapped to r in (0, inf): r = rm (1 + x) / (1 - x)
    rm = 1.0
    r = rm * (1 + xk) / (1 - xk)
    wr = (np.pi / (nrad + 1)) * np.sin(i_ * np.pi / (nrad + 1)) ** 2 * 2 * rm / (1 - xk) ** 2 / np.sqrt(1 - xk**2) * r**2
    ang, wa = _octahedral50()
    pts, wts, owner = [], [], []
    for a in range(na):
        p = R[a] + (r[:, None, None] * ang[None, :, :]).reshape(-1, 3)
        pts.append(p); wts.append((wr[:, None] * wa[None, :] * 4 * np.pi).ravel()); owner.append(np.full(len(p), a))
    pts, wts, owner = np.concatenate(pts), np.concatenate(wts), np.concatenate(owner)
    d = np.linalg.norm(pts[:, None, :] - R[None, :, :], axis=2)  # (npts, natom)
    cell = np.ones((len(pts), na))
    for a in range(na):
        for b in range(na):
            if a == b:
                continue
            mu = (d[:, a] - d[:, b]) / np.linalg.norm(R[a] - R[b])
            for _ in range(3):
                mu = 1.5 * mu - 0.5 * mu**3
            cell[:, a] *= 0.5 * (1 - mu)
    w = wts * cell[np.arange(len(pts)), owner] / cell.sum(axis=1)
    # ---- AO values and Laplacians on the grid (same normalised shell tables as the device)
    t = basis_tables(mol)
    nao = t["nao"]
    val, lap = np.zeros((len(pts), nao)), np.zeros((len(pts), nao))
    for sh in range(len(t["shell_l"])):
        l, ia, off = int(t["shell_l"][sh]), int(t["shell_atom"][sh]), int(t["shell_ao_off"][sh])
        if l > 2:
            raise NotImplementedError("model_mf: shells up to d")
        v = pts - R[ia]
        r2 = np.sum(v * v, axis=1)
        pe, pc = t["prim_exp"][t["shell_prim_off"][sh] : t["shell_prim_off"][sh + 1]], t["prim_coef"][t["shell_prim_off"][sh] : t["shell_prim_off"][sh + 1]]
        e = np.exp(-r2[:, None] * pe[None, :]) * pc[None, :]
        Rr, dRs, lapR = e.sum(1), -2.0 * (e * pe).sum(1), (e * 2 * pe * (2 * pe * r2[:, None] - 3)).sum(1)
        x, y, z = v.T
        if l == 0:
            S, dS = [0.28209479177387814 + 0 * x], [np.zeros_like(v)]
        elif l == 1:
            c = 0.4886025119029199
            S = [c * x, c * y, c * z]
            dS = [np.tile([c, 0, 0], (len(x), 1)), np.tile([0, c, 0], (len(x), 1)), np.tile([0, 0, c], (len(x), 1))]
        else:
            a_, b_, c_ = 1.0925484305920792, 0.31539156525252005, 0.5462742152960396
            zero = np.zeros_like(x)
            S = [a_ * x * y, a_ * y * z, b_ * (2 * z * z - x * x - y * y), a_ * x * z, c_ * (x * x - y * y)]
            dS = [np.stack([a_ * y, a_ * x, zero], 1), np.stack([zero, a_ * z, a_ * y], 1), np.stack([-2 * b_ * x, -2 * b_ * y, 4 * b_ * z], 1),
                  np.stack([a_ * z, zero, a_ * x], 1), np.stack([2 * c_ * x, -2 * c_ * y, zero], 1)]
        for m in range(2 * l + 1):
            val[:, off + m] = S[m] * Rr
            lap[:, off + m] = S[m] * lapR + 2.0 * dRs * np.sum(dS[m] * v, axis=1)  # lap S = 0
    # ---- model potential
    V = np.zeros(len(pts))
    for a in range(na):
        sym = mol.atom_pure_symbol(a)
        V -= screen.get(sym, float(mol.atom_charges()[a])) / d[:, a]
        if sym in mol._ecp:
            for l, terms in mol._ecp[sym][1]:
                if int(l) == -1:
                    for n, expand in enumerate(terms):
                        for al, c in expand:
                            V += c * d[:, a] ** (n - 2) * np.exp(-al * d[:, a] ** 2)
    wv = w[:, None] * val
    Smat = val.T @ wv
    Hmat = -0.5 * (wv.T @ lap) + val.T @ (wv * V[:, None])
    Hmat = 0.5 * (Hmat + Hmat.T)
    ev, L = np.linalg.eigh(Smat)
    X = L / np.sqrt(ev)  # S^-1/2 (canonical)
    eps, C = np.linalg.eigh(X.T @ Hmat @ X)
    C = X @ C
    mo = np.stack([C, C])
    occ = np.zeros((2, nao))
    for s in range(2):
        occ[s, : mol.nelec[s]] = 1.0
    mf = MeanField(mo, occ)
    mf.mo_energy = eps
    return mf


def _octahedral50():
    """50-point octahedral (Lebedev) rule, exact to l = 11: 6 vertices, 12 edge midpoints, 8 + 24 interior points
    (weights 4/315, 64/2835, 27/1280, 14641/725760 of the sphere; the reference's ECP grids, eval_ecp.py:278-336, stop at the same rule)."""
    pts, w = [], []
    for i in range(3):
        for s in (1, -1):
            p = [0.0, 0.0, 0.0]; p[i] = s
            pts.append(p); w.append(4.0 / 315.0)
    for i in range(3):
        for j in range(i + 1, 3):
            for si in (1, -1):
                for sj in (1, -1):
                    p = [0.0, 0.0, 0.0]; p[i] = si / np.sqrt(2); p[j] = sj / np.sqrt(2)
                    pts.append(p); w.append(64.0 / 2835.0)
    for sx in (1, -1):
        for sy in (1, -1):
            for sz in (1, -1):
                pts.append([sx / np.sqrt(3), sy / np.sqrt(3), sz / np.sqrt(3)]); w.append(27.0 / 1280.0)
    a, b = 1.0 / np.sqrt(11.0), 3.0 / np.sqrt(11.0)
    for perm in ((a, a, b), (a, b, a), (b, a, a)):
        for sx in (1, -1):
            for sy in (1, -1):
                for sz in (1, -1):
                    pts.append([sx * perm[0], sy * perm[1], sz * perm[2]]); w.append(14641.0 / 725760.0)
    return np.array(pts), np.array(w)
