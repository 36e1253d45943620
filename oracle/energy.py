"""Oracle: local-energy pieces (test infrastructure).

Kinetic + open-boundary Coulomb: ``pyqmc/observables/energy.py:19-65``.
ECP (semi-local pseudopotential) integrator: ``pyqmc/observables/eval_ecp.py``
— ``ecp`` :21-40, ``ecp_ea`` :83-132, ``ecp_mask`` :135-146, radial ``rnExp`` :182-200,
Legendre ``P_l`` :203-225, quadrature weights/points ``get_P_l`` :228-252,
``get_rot`` :255-275, grids :278-336 (the 6/18/26/50-point octahedral and 12/32-point
icosahedral rules of Mitas, Shirley & Ceperley, JCP 95, 3467 (1991)).
Harness: ``EnergyAccumulator.__call__`` ``pyqmc/observables/accumulators.py:60-75``.

Randomness is injected, never drawn here: ``rot`` is one 3x3 rotation per
(electron, ECP atom) call and ``unif`` one uniform per (electron, ECP atom, walker),
consumed in the reference's call order (electron-major, atoms in ``mol._atom`` order).
"""

import numpy as np


def kinetic(configs, wf):
    """energy.py:57-65."""
    W, N, _ = configs.configs.shape
    ke, grad2 = np.zeros(W), np.zeros(W)
    for e in range(N):
        grad, lap = wf.gradient_laplacian(e, configs.electron(e))
        ke += -0.5 * np.real(lap)
        grad2 += np.sum(np.abs(grad) ** 2, axis=0)
    return ke, grad2


def coulomb(mol, configs):
    """energy.py:28-54: (ee (W,), ei (W,), ii scalar)."""
    x = configs.configs
    N = x.shape[1]
    iu, ju = np.triu_indices(N, k=1)
    ee = np.sum(1.0 / np.linalg.norm(x[:, iu] - x[:, ju], axis=-1), axis=1) if N > 1 else np.zeros(len(x))
    ei = np.zeros(len(x))
    R, Z = np.asarray(mol.atom_coords()), np.asarray(mol.atom_charges())
    for c, coord in zip(Z, R):
        ei += -c * np.sum(1.0 / np.linalg.norm(x - coord, axis=2), axis=1)
    ii = 0.0
    for i in range(len(R)):
        for j in range(i + 1, len(R)):
            ii += Z[i] * Z[j] / np.linalg.norm(R[i] - R[j])
    return ee, ei, ii


# ---------------------------------------------------------------- ECP
def ecp_channels(ecp_entry):
    """eval_ecp.py:160-200.  Returns list ordered like the reference's v_l columns:
    non-local channels l=0,1,... first, local channel (key -1) LAST (column index -1).
    Each channel = (n (k,), exps (k,), coefs (k,)) for sum c r^n e^{-a r^2}, n = index-2."""
    chans = {}
    for l, terms in ecp_entry[1]:
        n, a, c = [], [], []
        for idx, expand in enumerate(terms):
            for line in expand:
                n.append(idx - 2)
                a.append(line[0])
                c.append(line[1])
        chans[int(l)] = (np.asarray(n, float), np.asarray(a, float), np.asarray(c, float))
    nl = len(chans)
    order = list(range(nl - 1)) + [-1]
    return [chans[l] for l in order]


def v_l(channels, r):
    out = np.zeros((len(r), len(channels)))
    for k, (n, a, c) in enumerate(channels):
        out[:, k] = np.sum(r[:, None] ** n * c * np.exp(-a * r[:, None] ** 2), axis=1)
    return out


def legendre(x, l):
    """eval_ecp.py:203-225 (l = -1 -> 0)."""
    if l == -1:
        return np.zeros_like(x)
    return [np.ones_like(x), x, 0.5 * (3 * x * x - 1), 0.5 * (5 * x**3 - 3 * x), 0.125 * (35 * x**4 - 30 * x * x + 3)][l]


def quadrature(naip):
    """eval_ecp.py:278-336: the six rules of Mitas, Shirley & Ceperley, JCP 95, 3467 (1991), points in the reference's order.
    Octahedral families: the 26 non-zero points of {-1,0,1}^3 (x slowest) split by how many coordinates are non-zero and
    normalised (A: 6 axes, B: 12 edge midpoints, C: 8 corners), D: the 24 points (+-1,+-1,+-3)/sqrt(11) as three cyclic column
    shifts.  Icosahedral families by polar angle: A poles, B at atan 2, C at the two angles c_1, c_2."""
    cube = np.array([[x, y, z] for x in (-1, 0, 1) for y in (-1, 0, 1) for z in (-1, 0, 1)], dtype=float)
    nnz = np.count_nonzero(cube, axis=1)
    oct_ = [cube[nnz == k] / np.sqrt(k) for k in (1, 2, 3)]
    d1 = oct_[2] * np.sqrt(3.0 / 11.0) * np.array([1.0, 1.0, 3.0])
    oct_.append(np.concatenate([np.roll(d1, i, axis=1) for i in range(3)]))
    k = np.arange(10)
    s5 = np.sqrt(5.0)
    b1, c1, c2 = np.arctan(2.0), np.arccos((2 + s5) / np.sqrt(15 + 6 * s5)), np.arccos(1 / np.sqrt(15 + 6 * s5))

    def sph(th, ph):
        return np.stack([np.sin(th) * np.cos(ph), np.sin(th) * np.sin(ph), np.cos(th)], axis=1)

    ico = [sph(np.array([0.0, np.pi]), np.zeros(2)), sph(np.tile([b1, np.pi - b1], 5), k * np.pi / 5),
           sph(np.concatenate([np.tile([np.pi - c1, c1], 5), np.tile([np.pi - c2, c2], 5)]), np.tile(k * np.pi / 5, 2))]
    rules = {6: (oct_[:1], [1 / 6]), 18: (oct_[:2], [1 / 30, 1 / 15]), 26: (oct_[:3], [1 / 21, 4 / 105, 27 / 840]),
             50: (oct_[:4], [4 / 315, 64 / 2835, 27 / 1280, 14641 / 725760]),
             12: (ico[:2], [1 / 12, 1 / 12]), 32: (ico[:3], [5 / 168, 5 / 168, 27 / 840])}
    if naip not in rules:
        raise ValueError(f"Possible AIPs are one of {sorted(rules)}")  # eval_ecp.py:266-267
    fams, w = rules[naip]
    return np.concatenate(fams), np.concatenate([np.full(len(f), wi) for f, wi in zip(fams, w)])


def ecp_ea(mol, configs, wf, e, atom_index, threshold, rot, unif, naip=None):
    """eval_ecp.py:83-132 for one (electron, atom).  rot (3,3); unif (W,)."""
    x = configs.configs
    W = x.shape[0]
    sym = mol.atom_pure_symbol(atom_index)
    apos = np.asarray(mol.atom_coords()[atom_index])
    channels = ecp_channels(mol._ecp[sym])
    nl = len(channels)
    if naip is None:
        naip = 6 if nl <= 2 else 12
    r_vec = x[:, e, :] - apos
    if hasattr(mol, "a"):  # configs.dist.dist_i is the minimal-image displacement in a periodic cell (eval_ecp.py:95)
        from .pbc import minimal_image

        r_vec = minimal_image(mol.lattice_vectors())(r_vec)
    r = np.linalg.norm(r_vec, axis=-1)
    v = v_l(channels, r)
    if threshold > 0:  # eval_ecp.py:135-146
        l = 2 * np.arange(nl - 1) + 1
        prob = np.minimum(1.0, np.abs(v[:, :-1]) @ (threshold * (2 * l + 1)))
    else:
        prob = np.ones(W)
    mask = prob > unif
    mv = v[mask].copy()
    mv[:, :-1] /= prob[mask, None]
    pts, wts = quadrature(naip)
    rot_vec = (rot @ pts.T).T  # eval_ecp.py:271
    rm, rvm = r[mask], r_vec[mask]
    r_i = rm[:, None, None] * rot_vec[None]  # (Wm, naip, 3)
    cos = np.einsum("ik,ijk->ij", rvm, r_i) / (rm[:, None] * np.linalg.norm(r_i, axis=-1))
    P = np.zeros((len(rm), naip, nl))
    for k in range(nl):  # column k holds l=k for non-local, last column l=-1 -> 0
        l = k if k < nl - 1 else -1
        P[:, :, k] = (2 * l + 1) * legendre(cos, l) * wts[None]
    epos = np.repeat(x[:, e, None, :], naip, axis=1)
    epos[mask] = (x[mask, e, :] - rvm)[:, None] + r_i
    val = np.zeros(W, dtype=getattr(wf, "dtype", float))  # eval_ecp.py:89
    if np.any(mask):
        ratio = wf.testvalue(e, configs.make_irreducible(e, epos, mask), mask)[0]
        val[mask] = np.einsum("ij,ik,ijk->i", ratio, mv, P)
    else:
        ratio = np.zeros((0, naip))
    val += v[:, -1]
    return {"total": val, "local": v[:, -1], "mask": mask, "prob": prob, "ratio": ratio, "epos": epos, "v_l": mv, "P_l": P}


def ecp_atoms(mol):
    return [i for i in range(mol.natm) if mol.atom_pure_symbol(i) in mol._ecp]


def ecp(mol, configs, wf, threshold, rot_tape, unif_tape, naip=None):
    """eval_ecp.py:21-40.  rot_tape (N, n_ecp_atoms, 3, 3); unif_tape (N, n_ecp_atoms, W)."""
    W, N = configs.configs.shape[:2]
    tot = np.zeros(W, dtype=getattr(wf, "dtype", float))
    for e in range(N):
        for k, ia in enumerate(ecp_atoms(mol)):
            tot += ecp_ea(mol, configs, wf, e, ia, threshold, rot_tape[e, k], unif_tape[e, k], naip)["total"]
    return tot


def energy(mol, configs, wf, threshold, rot_tape, unif_tape, naip=None, ewald_kws=None):
    """accumulators.py:60-75 (Ewald for a periodic cell, :52-53)."""
    if hasattr(mol, "a"):
        from .pbc import Ewald

        ee, ei, ii = Ewald(mol, **(ewald_kws or {})).energy(configs)
    else:
        ee, ei, ii = coulomb(mol, configs)
    ecp_val = ecp(mol, configs, wf, threshold, rot_tape, unif_tape, naip) if mol._ecp else np.zeros(len(ee))
    ke, grad2 = kinetic(configs, wf)
    return {"ke": ke, "ee": ee, "ei": ei, "ecp": ecp_val, "grad2": grad2, "total": ke + ee + ei + ecp_val + ii}
