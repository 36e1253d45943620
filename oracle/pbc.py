"""TEST INFRASTRUCTURE — CPU restatement of the reference's periodic-boundary pieces (numpy, fp64).

* ``enforce_pbc``            pyqmc/pbc/pbc.py:18-49
* ``minimal_image``          pyqmc/configurations/distance.py:83-159 (diagonal / orthogonal / general rules)

Pinned by tests/golden/g13_pbc.npz (outputs of the real reference, tests/golden/make_golden.py:g_pbc), which also
holds the known-answer table of the reference's own tests/unit/test_pbcs.py:19-72.
"""

import numpy as np


def enforce_pbc(lattvecs, epos):
    frac = np.einsum("...ij,jk->...ik", epos, np.linalg.inv(lattvecs))
    wrap, rem = np.divmod(frac, 1)
    return np.dot(rem, lattvecs), wrap


def lattice_kind(latvec, tol=1e-10):
    def diag(m):
        return np.all(np.abs(m - np.diag(np.diagonal(m))) < tol)

    if diag(latvec):
        return "diagonal"
    return "orthogonal" if diag(latvec @ latvec.T) else "general"


def minimal_image(latvec):
    """Returns f(d) mapping displacement arrays (...,3) to their nearest-image representatives."""
    latvec = np.asarray(latvec, dtype=float)
    kind = lattice_kind(latvec)
    inv = np.linalg.inv(latvec)
    grid = np.meshgrid(*[np.arange(3)] * 3)
    shifts = (np.stack([g.ravel() for g in grid], axis=0).T - 1) @ latvec  # distance.py:113-118

    def f(d):
        d = np.array(d, dtype=float)
        if kind == "diagonal":
            for i in range(3):
                L = latvec[i, i]
                d[..., i] = (d[..., i] + L / 2) % L - L / 2
            return d
        if kind == "orthogonal":
            return (((d @ inv) + 0.5) % 1 - 0.5) @ latvec
        cand = d[None] + shifts.reshape((-1,) + (1,) * (d.ndim - 1) + (3,))
        best = np.argmin(np.sum(cand**2, axis=-1), axis=0)
        return np.take_along_axis(cand, best[None, ..., None], axis=0)[0]

    return f


# ------------------------------------------------------------------------------------ periodic orbitals
# Restatement of the reference's in-repo periodic GTO evaluator and k-point MO evaluator:
#   cut-offs            max_Ls                         pyqmc/wf/numba/pbcgto.py:549-591
#   lattice-summed AOs  _pbc_eval_gto(_grad,_lap)      pbcgto.py:99-506   (r2 > atom cut: skip image; per-shell cut)
#   Bloch phases        phases = exp(i Ls.k)           pbcgto.py:620
#   k-point wrapper     PBCOrbitalEvaluatorKpoints     pyqmc/wf/orbitals.py:118-255 (fold into the primitive cell,
#                       wrap phase (-1)^round(k.R/pi) :34-35,203-213, per-k MO blocks :221-239)
# Pinned by tests/golden/g15_pbc_orbitals.npz (outputs of those reference functions, make_golden.py:g_pbc_slater).
# Only real Bloch phases (k-points with e^{ik.L} = +-1) and zero supercell twist are restated.


def max_distance_in_cell(lvecs):
    combos = np.array([[1.0, 1.0, 1.0], [-1.0, 1.0, 1.0], [1.0, -1.0, 1.0], [1.0, 1.0, -1.0]])
    vecs = combos @ lvecs
    return vecs[np.argmax(np.sum(vecs**2, axis=-1))] / 2


def gto_cutoffs(table, Ls, lvecs, expcutoff):
    """(num_Ls per atom, r^2 cut per atom, r^2 cut per shell) — pbcgto.py:549-591.  ``table``: oracle.gto.AOTable."""
    natom = len(table.coords)
    v = max_distance_in_cell(lvecs)
    r2 = np.sum((v - Ls) ** 2, axis=-1)
    num, acut, lcut = np.ones(natom, dtype=int), np.zeros(natom), np.zeros(len(table.shells))
    num[:] = 0
    for i, (ia, l, exps, coefs, off) in enumerate(table.shells):
        log_c = np.log(np.abs(coefs))
        if l == 0:
            lcut[i] = np.amax((expcutoff + log_c) / exps)
        else:
            lconst = 0.5 * np.log(0.5 * l / np.amin(exps)) * l
            lcut[i] = np.amax((expcutoff + log_c + lconst) / exps)
        acut[ia] = max(acut[ia], lcut[i])
        with np.errstate(divide="ignore"):
            min_exp = np.amin(exps[None, :] * r2[:, None] - log_c[None, :] - 0.5 * np.log(r2)[:, None] * l, axis=1)
        where = np.where(min_exp < expcutoff)[0]
        num[ia] = max(num[ia], where.max() + 1 if len(where) else 1)
    return num, acut, lcut


class PeriodicAOTable:
    """Primitive-cell AO tables + sorted lattice translations + cut-offs (PeriodicAtomicOrbitalEvaluator,
    pbcgto.py:594-636).  ``Ls`` must already be sorted by norm (:603)."""

    def __init__(self, cell, kpts, Ls, precision=1e-2):
        from . import gto

        self.table = gto.AOTable(cell)
        self.kpts = np.asarray(kpts, dtype=float).reshape(-1, 3)
        self.Ls = np.asarray(Ls, dtype=float)
        self.num_Ls, self.atom_cut, self.shell_cut = gto_cutoffs(self.table, self.Ls, cell.lattice_vectors(),
                                                                 -3.5 * np.log(precision))
        ph = np.exp(1j * self.Ls @ self.kpts.T)
        self.phases = ph.real if np.abs(ph.imag).max() < 1e-9 else ph  # np.real_if_close, pbcgto.py:620-621


def _eval_ao_pbc_c(pt, pts, ncomp):
    """oracle/ao_eval.c:ao_eval_pbc — the compiled twin of the routine below (same tests, same image order, same sums)."""
    import ctypes

    from . import gto

    f = gto._flat_tables(pt.table)
    cplx = np.iscomplexobj(pt.phases)
    nk = len(pt.kpts)
    out = np.zeros((nk, ncomp, len(pts), pt.table.nao), dtype=complex if cplx else float)
    ph = np.ascontiguousarray(pt.phases)
    dp, ip = ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int)
    c = lambda a, ty: np.ascontiguousarray(a, dtype=ty)
    Ls, numL, acut, scut = c(pt.Ls, float), c(pt.num_Ls, np.int32), c(pt.atom_cut, float), c(pt.shell_cut, float)
    rc = gto._ao_lib().ao_eval_pbc(ncomp, len(pts), pts.ctypes.data_as(dp), len(f["l"]), f["atom"].ctypes.data_as(ip), f["l"].ctypes.data_as(ip),
                                   f["prim_off"].ctypes.data_as(ip), f["exps"].ctypes.data_as(dp), f["coefs"].ctypes.data_as(dp),
                                   f["ao_off"].ctypes.data_as(ip), f["xyz"].ctypes.data_as(dp), pt.table.nao, nk, Ls.ctypes.data_as(dp),
                                   numL.ctypes.data_as(ip), acut.ctypes.data_as(dp), scut.ctypes.data_as(dp),
                                   ph.view(float).ctypes.data_as(dp) if cplx else ph.ctypes.data_as(dp), int(cplx), out.view(float).ctypes.data_as(dp))
    return out if rc == 0 else None


def eval_ao_pbc(pt, pts, ncomp):
    """(nk, ncomp, npts, nao) lattice-summed AOs at points inside the primitive cell."""
    from . import gto

    t = pt.table
    pts = np.ascontiguousarray(np.asarray(pts, dtype=float).reshape(-1, 3))
    if gto._AO_BACKEND == "c" and ncomp in (1, 4, 5) and len(pts):
        out = _eval_ao_pbc_c(pt, pts, ncomp)  # None: a shell with l > 3
        if out is not None:
            return out
    out = np.zeros((len(pt.kpts), ncomp, len(pts), t.nao), dtype=pt.phases.dtype)
    deriv = ncomp > 1
    for ia in range(len(t.coords)):
        shells = [(i, s) for i, s in enumerate(t.shells) if s[0] == ia]
        for j in range(pt.num_Ls[ia]):
            v = pts - t.coords[ia] - pt.Ls[j]
            r2 = np.sum(v * v, axis=1)
            keep = ~(r2 > pt.atom_cut[ia])
            if not keep.any():
                continue
            vk, r2k = v[keep], r2[keep]
            S, dS = gto.solid_harmonics(vk, t.max_l, deriv)
            for i, (_, l, exps, coefs, off) in shells:
                sel = r2k < pt.shell_cut[i] if ncomp == 1 else ~(r2k > pt.shell_cut[i])  # pbcgto.py:254 vs :356
                if not sel.any():
                    continue
                prim = np.exp(-r2k[sel, None] * exps[None, :]) * coefs[None, :]
                R = prim.sum(axis=1)
                sl = slice(l * l, (l + 1) * (l + 1))
                val = np.zeros((ncomp, sel.sum(), 2 * l + 1))
                val[0] = S[sel][:, sl] * R[:, None]
                if deriv:
                    dR = -(2.0 * (prim * exps[None, :]).sum(axis=1))[:, None] * vk[sel]
                    for c in range(3):
                        val[1 + c] = dS[sel][:, sl, c] * R[:, None] + S[sel][:, sl] * dR[:, c : c + 1]
                if ncomp == 5:
                    lapR = (prim * (2.0 * exps[None, :]) * (2.0 * exps[None, :] * r2k[sel, None] - 3.0)).sum(axis=1)
                    val[4] = S[sel][:, sl] * lapR[:, None] + 2.0 * np.einsum("pmi,pi->pm", dS[sel][:, sl, :], dR)
                idx = np.nonzero(keep)[0][sel]
                for k in range(len(pt.kpts)):
                    out[k][:, idx, off : off + 2 * l + 1] += pt.phases[j, k] * val
    return out


class PeriodicOrbitals:
    """MO evaluator for a supercell built from k-points of a primitive cell (orbitals.py:118-255).
    mo_coeff[s][k]: (nao_prim, nmo_k); MO columns of a spin are the k blocks concatenated (:154-160)."""

    def __init__(self, supercell, kpts, mo_coeff, Ls, precision=1e-2):
        self.prim = supercell.original_cell
        self.S = np.asarray(supercell.S, dtype=float)
        self.Lprim = self.prim.lattice_vectors()
        self.aotab = PeriodicAOTable(self.prim, kpts, Ls, precision)
        self.kpts = self.aotab.kpts
        self.mo = [[np.asarray(m) for m in mo_coeff[s]] for s in (0, 1)]
        self.complex = np.iscomplexobj(self.aotab.phases) or any(np.iscomplexobj(m) for sp in self.mo for m in sp)

    def aos(self, pts, ncomp):
        """pts: TRUE (unfolded) positions.  Folding them into the supercell gives the container's wrap counters, folding
        again into the primitive cell ``primwrap``; the wrap phase is exp(i k . (wrap @ S + primwrap) @ Lprim)
        (orbitals.py:199-213; the real form (-1)^round(k.R/pi) :34-35 when nothing is complex)."""
        import time

        from . import gto as _gto

        t0 = time.perf_counter()
        pts = np.asarray(pts, dtype=float).reshape(-1, 3)
        cell_pts, wrap_s = enforce_pbc(self.S @ self.Lprim, pts)
        prim_pts, primwrap = enforce_pbc(self.Lprim, cell_pts)
        ao = eval_ao_pbc(self.aotab, prim_pts, ncomp)
        _gto.AO_SECONDS += time.perf_counter() - t0  # (the CPU baselines report the AO share of their wall time)
        wrap = wrap_s @ self.S + primwrap
        kdotR = self.kpts @ self.Lprim.T @ wrap.T  # (nk, npts)
        wrap_phase = np.exp(1j * kdotR) if self.complex else (-1.0) ** np.round(kdotR / np.pi)
        return ao * wrap_phase[:, None, :, None]

    def mos(self, ao, s):
        return np.concatenate([ao[k] @ self.mo[s][k] for k in range(len(self.kpts))], axis=-1)


# ------------------------------------------------------------------------------------ Ewald
class Ewald:
    """Coulomb energy of a periodic cell — restatement of ``pyqmc/observables/ewald.py`` (alpha and reciprocal
    vectors :125-148, constants :150-190, ion-ion :192-238, electron sums :240-304, total :330-354).
    Pinned by tests/golden/g16_pbc_energy.npz."""

    def __init__(self, cell, ewald_gmax=200, nlatvec=1):
        from math import erfc as _erfc

        self._erfc = np.vectorize(_erfc)
        self.charges = np.asarray(cell.atom_charges(), dtype=float)
        self.coords = np.asarray(cell.atom_coords(), dtype=float)
        self.latvec = np.asarray(cell.lattice_vectors(), dtype=float)
        xyz = np.stack(np.meshgrid(*[np.arange(-nlatvec, nlatvec + 1)] * 3, indexing="ij"), axis=-1).reshape(-1, 3)
        self.disp = xyz @ self.latvec
        vol = np.linalg.det(self.latvec)
        recvec = np.linalg.inv(self.latvec).T
        self.alpha = 5.0 / np.amin(1 / np.linalg.norm(recvec, axis=1))
        # positive half space, bounded so that nothing with weight > 1e-10 is lost (|G| <= 12 alpha is far beyond it)
        nmax = np.minimum(np.ceil(12 * self.alpha * np.linalg.norm(self.latvec, axis=1) / (2 * np.pi)).astype(int), ewald_gmax)
        idx = [np.mgrid[1 : nmax[0] + 1, -nmax[1] : nmax[1] + 1, -nmax[2] : nmax[2] + 1].reshape(3, -1),
               np.mgrid[0:1, 1 : nmax[1] + 1, -nmax[2] : nmax[2] + 1].reshape(3, -1),
               np.mgrid[0:1, 0:1, 1 : nmax[2] + 1].reshape(3, -1)]
        g = np.concatenate(idx, axis=1).T @ (recvec * 2 * np.pi)
        g2 = np.sum(g * g, axis=1)
        wt = 4 * np.pi * np.exp(-g2 / (4 * self.alpha**2)) / (vol * g2)
        self.g, self.gw = g[wt > 1e-10], wt[wt > 1e-10]
        self.i_sum = self.charges.sum()
        ii_sum2 = np.sum(self.charges**2)
        self.ijconst = -np.pi / (vol * self.alpha**2)
        self.squareconst = -self.alpha / np.sqrt(np.pi) + self.ijconst / 2
        self.ii_const = (self.i_sum**2 - ii_sum2) / 2 * self.ijconst + ii_sum2 * self.squareconst
        self.mi = minimal_image(self.latvec)
        if len(self.charges) > 1:
            iu, ju = np.triu_indices(len(self.charges), k=1)
            d = self.mi(self.coords[iu] - self.coords[ju])
            real = np.sum((self.charges[iu] * self.charges[ju])[:, None] * self._cij(d[:, None, :] + self.disp[None]))
        else:
            real = 0.0
        self.ion_exp = np.exp(1j * self.g @ self.coords.T) @ self.charges
        self.ion_ion = real + self.gw @ np.abs(self.ion_exp) ** 2

    def _cij(self, rvec):
        r = np.linalg.norm(rvec, axis=-1)
        return self._erfc(self.alpha * r) / r

    def energy(self, configs):
        x = configs.configs
        W, N = x.shape[:2]
        d_ei = self.mi(x[:, None, :, :] - self.coords[None, :, None, :])  # (W, atom, elec, 3)
        ei = -np.einsum("a,wae->w", self.charges, self._cij(d_ei[..., None, :] + self.disp).sum(axis=-1))
        ee = np.zeros(W)
        if N > 1:
            iu, ju = np.triu_indices(N, k=1)
            d = self.mi(x[:, iu] - x[:, ju])
            ee = self._cij(d[..., None, :] + self.disp).sum(axis=(-1, -2))
        gr = np.einsum("wik,jk->wij", x, self.g)
        ssin, scos = np.sin(gr).sum(axis=1), np.cos(gr).sum(axis=1)
        ee = ee + (ssin**2 + scos**2) @ self.gw + N * (N - 1) / 2 * self.ijconst + N * self.squareconst
        ei = ei + 2 * ((-self.ion_exp.real * scos - self.ion_exp.imag * ssin) @ self.gw) - N * self.i_sum * self.ijconst
        return ee, ei, self.ion_ion + self.ii_const
