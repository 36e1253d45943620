"""Oracle: molecular Gaussian-type atomic orbitals (test infrastructure).

Specification followed: ``pyqmc/wf/numba/gto.py`` —
normalisation ``normalize_basis_coeffs`` :375-405, table construction
``AtomicOrbitalEvaluator.__init__`` :435-470, value ``mol_eval_gto`` :89-136,
gradient ``mol_eval_gto_grad`` :139-194, Laplacian ``mol_eval_gto_lap`` :197-254,
radial parts :257-321; real solid harmonics ``numba/spherical_harmonics.py:40-200``
(orthonormal real Y_lm times r^l; the l=1 triple is ordered x,y,z, :57-66).
AO->MO contraction: ``pyqmc/wf/orbitals.py:95-96``.

Written vectorised over points (the reference loops points inside each atom).
"""

import math

import numpy as np
from scipy.special import gamma as _gamma

_PI = math.pi
_S0 = 0.5 / math.sqrt(_PI)
_P1 = math.sqrt(3.0 / (4.0 * _PI))
_D_XY = 0.5 * math.sqrt(15.0 / _PI)
_D_Z2 = 0.25 * math.sqrt(5.0 / _PI)
_D_X2Y2 = 0.25 * math.sqrt(15.0 / _PI)
_F_3 = 0.25 * math.sqrt(35.0 / (2.0 * _PI))
_F_2 = 0.5 * math.sqrt(105.0 / _PI)
_F_1 = 0.25 * math.sqrt(21.0 / (2.0 * _PI))
_F_0 = 0.25 * math.sqrt(7.0 / _PI)
_F_2C = 0.25 * math.sqrt(105.0 / _PI)
LMAX = 3


def solid_harmonics(v, lmax, deriv):
    """Real solid harmonics S_lm(v) for l<=lmax at points v (...,3).

    Returns S (...,(lmax+1)^2) and, if ``deriv``, dS (...,(lmax+1)^2,3).
    Index l*l+m' as in spherical_harmonics.py (l=1 block ordered x,y,z).
    """
    if lmax > LMAX:
        raise NotImplementedError("oracle solid harmonics are written out for l<=3")
    x, y, z = v[..., 0], v[..., 1], v[..., 2]
    n = (lmax + 1) ** 2
    S = np.zeros(v.shape[:-1] + (n,))
    dS = np.zeros(v.shape[:-1] + (n, 3)) if deriv else None
    S[..., 0] = _S0
    if lmax >= 1:
        for i in range(3):
            S[..., 1 + i] = _P1 * v[..., i]
            if deriv:
                dS[..., 1 + i, i] = _P1
    if lmax >= 2:
        S[..., 4] = _D_XY * x * y
        S[..., 5] = _D_XY * y * z
        S[..., 6] = _D_Z2 * (2 * z * z - x * x - y * y)
        S[..., 7] = _D_XY * x * z
        S[..., 8] = _D_X2Y2 * (x * x - y * y)
        if deriv:
            dS[..., 4, 0], dS[..., 4, 1] = _D_XY * y, _D_XY * x
            dS[..., 5, 1], dS[..., 5, 2] = _D_XY * z, _D_XY * y
            dS[..., 6, 0], dS[..., 6, 1], dS[..., 6, 2] = -2 * _D_Z2 * x, -2 * _D_Z2 * y, 4 * _D_Z2 * z
            dS[..., 7, 0], dS[..., 7, 2] = _D_XY * z, _D_XY * x
            dS[..., 8, 0], dS[..., 8, 1] = 2 * _D_X2Y2 * x, -2 * _D_X2Y2 * y
    if lmax >= 3:
        x2, y2, z2 = x * x, y * y, z * z
        S[..., 9] = _F_3 * y * (3 * x2 - y2)
        S[..., 10] = _F_2 * x * y * z
        S[..., 11] = _F_1 * y * (4 * z2 - x2 - y2)
        S[..., 12] = _F_0 * z * (2 * z2 - 3 * x2 - 3 * y2)
        S[..., 13] = _F_1 * x * (4 * z2 - x2 - y2)
        S[..., 14] = _F_2C * z * (x2 - y2)
        S[..., 15] = _F_3 * x * (x2 - 3 * y2)
        if deriv:
            dS[..., 9, 0], dS[..., 9, 1] = _F_3 * 6 * x * y, _F_3 * (3 * x2 - 3 * y2)
            dS[..., 10, 0], dS[..., 10, 1], dS[..., 10, 2] = _F_2 * y * z, _F_2 * x * z, _F_2 * x * y
            dS[..., 11, 0] = _F_1 * (-2 * x * y)
            dS[..., 11, 1] = _F_1 * (4 * z2 - x2 - 3 * y2)
            dS[..., 11, 2] = _F_1 * 8 * y * z
            dS[..., 12, 0] = _F_0 * (-6 * x * z)
            dS[..., 12, 1] = _F_0 * (-6 * y * z)
            dS[..., 12, 2] = _F_0 * (6 * z2 - 3 * x2 - 3 * y2)
            dS[..., 13, 0] = _F_1 * (4 * z2 - 3 * x2 - y2)
            dS[..., 13, 1] = _F_1 * (-2 * x * y)
            dS[..., 13, 2] = _F_1 * 8 * x * z
            dS[..., 14, 0], dS[..., 14, 1], dS[..., 14, 2] = _F_2C * 2 * x * z, -_F_2C * 2 * y * z, _F_2C * (x2 - y2)
            dS[..., 15, 0], dS[..., 15, 1] = _F_3 * (3 * x2 - 3 * y2), -_F_3 * 6 * x * y
    return S, dS


def normalized_contraction(l, prims):
    """Primitive coefficients of one contracted shell after PySCF ``gto_norm``-style
    normalisation (numba/gto.py:375-405).  prims: (nprim,2) [exponent, raw coef]."""
    prims = np.asarray(prims, dtype=float)
    a, c = prims[:, 0], prims[:, 1]
    m = l + 1.5
    gm = _gamma(m)
    cs = c * np.sqrt(2.0 * (2.0 * a) ** m / gm)  # per-primitive radial normalisation
    overlap = gm / (2.0 * (a[:, None] + a[None, :]) ** m)
    return cs / math.sqrt(cs @ overlap @ cs)


class AOTable:
    """Flat shell tables for a molecule (numba/gto.py:435-470).

    shells: list of (atom_index, l, exps (nprim,), coefs (nprim,), ao_offset)
    """

    def __init__(self, mol):
        self.coords = np.asarray(mol.atom_coords(), dtype=float)
        self.shells = []
        off = 0
        for ia in range(len(self.coords)):
            sym = mol.atom_pure_symbol(ia)
            for sh in mol._basis[sym]:
                l = int(sh[0])
                prims = np.asarray(sh[1:], dtype=float)
                self.shells.append((ia, l, prims[:, 0].copy(), normalized_contraction(l, prims), off))
                off += 2 * l + 1
        self.nao = off
        self.max_l = max(s[1] for s in self.shells)


def eval_ao(table, pts, ncomp):
    """AO values (ncomp=1), +gradient (ncomp=4), +Laplacian (ncomp=5).

    Returns (ncomp, npts, nao) — the layout of ``mol_eval_gto*`` after the final
    transpose (gto.py:136,194,254).  chi = S_lm(r) R(r^2); grad = dS R + S dR with
    dR_i = -2 a x_i c e^{-a r^2} (:290-297); lap = S sum 2a(2a r^2-3) c e^{-a r^2}
    + 2 grad S . grad R (:241-250, :313-321; uses lap S = 0)."""
    pts = np.asarray(pts, dtype=float).reshape(-1, 3)
    out = np.zeros((ncomp, pts.shape[0], table.nao))
    deriv = ncomp > 1
    cache = {}
    for ia, l, exps, coefs, off in table.shells:
        if ia not in cache:
            v = pts - table.coords[ia]
            r2 = np.sum(v * v, axis=1)
            S, dS = solid_harmonics(v, table.max_l, deriv)
            cache = {ia: (v, r2, S, dS)}
        v, r2, S, dS = cache[ia]
        prim = np.exp(-r2[:, None] * exps[None, :]) * coefs[None, :]  # (npts,nprim)
        R = prim.sum(axis=1)
        sl = slice(l * l, (l + 1) * (l + 1))
        ao = slice(off, off + 2 * l + 1)
        out[0, :, ao] = S[:, sl] * R[:, None]
        if deriv:
            dRs = -(2.0 * (prim * exps[None, :]).sum(axis=1))  # dR/dx_i = dRs * x_i
            dR = dRs[:, None] * v
            for i in range(3):
                out[1 + i, :, ao] = dS[:, sl, i] * R[:, None] + S[:, sl] * dR[:, i : i + 1]
        if ncomp == 5:
            lapR = (prim * (2.0 * exps[None, :]) * (2.0 * exps[None, :] * r2[:, None] - 3.0)).sum(axis=1)
            out[4, :, ao] = S[:, sl] * lapR[:, None] + 2.0 * np.einsum("pmi,pi->pm", dS[:, sl, :], dR)
    return out


def eval_mo(ao, mo_coeff):
    """orbitals.py:95-96: ``ao[0].dot(C_s)`` for every component/point."""
    return ao @ mo_coeff
