"""Two-body density matrix accumulator — SURVEY.md §8(f3); public interface of ``pyqmc/observables/tbdm.py``.

One spin sector ``tbdm[s1,s2][i,j,k,l] = <c+_{s1,i} c+_{s2,k} c_{s2,l} c_{s1,j}>`` (PySCF's index convention), sampled
by moving an electron pair (a, b) to auxiliary positions (r1', r2') drawn from the orbital densities (Eq. 10 of
DOI:10.1063/1.4793531; tbdm.py:188-283).  On the device handle of an ``obdm.OrbitalEvaluator``:

* two resident auxiliary walks (``pqa_dm_walk``, slots 0 and 1) and the basis orbitals at the electrons of both groups
  (``pqa_dm_points``);
* the estimator in factorised form (``pqa_tbdm_accumulate`` -> ``k_tbdm_acc``): with the pair ratios
  ``R[a][b] = Psi(r_a -> r1', r_b -> r2') / Psi`` the sum over electron pairs is ``M = Phi_a^T R Phi_b`` — two small
  matrix products per configuration — and ``value[(i,j,k,l)] = M[i][k] conj(phi_j(r1')) conj(phi_l(r2')) / (f1 f2)``.
  The reference builds a (configurations x pairs x index tuples) tensor instead.

The pair ratios come from the wave function's protocol: ``testvalue`` + ``updateinternals`` move electron a to r1'
(Sherman-Morrison on the device), ``testvalue_many`` gives the ratios of all partners b at once (``k_testvalue_many``),
and a second ``updateinternals`` moves a back.  ``numpy.random`` is consumed in the reference's order.
"""

import numpy as np

from . import _ffi
from .obdm import AuxiliaryWalkers, OrbitalEvaluator


class TBDMAccumulator:
    """Keys ``value`` (M,), ``norm_a`` (norb_a,), ``norm_b`` (norb_b,) per configuration for the M index tuples ``ijkl``
    (default: the whole sector).  ``orb_coeff`` (2, nao, norb): orbital basis per spin; ``spin`` = (s1, s2)."""

    def __init__(self, mol, orb_coeff, spin, nsweeps=4, tstep=0.50, warmup=200, naux=None, ijkl=None, kpts=None,
                 eval_gto_precision=None, device=0):
        self.orbitals = OrbitalEvaluator(mol, orb_coeff, kpts=kpts, eval_gto_precision=eval_gto_precision, device=device)
        self._mol, self.dtype = self.orbitals.mol, self.orbitals.mo_dtype
        self._tstep, self._nsweeps, self._warmup, self._naux, self._spin_sector = tstep, nsweeps, warmup, naux, tuple(spin)
        nup, ndn = self._mol.nelec
        self._electrons = [np.arange(s * nup, nup + s * ndn) for s in spin]
        na, nb = self.orbitals.nmo()
        if ijkl is None:
            ijkl = np.stack(np.meshgrid(np.arange(na), np.arange(na), np.arange(nb), np.arange(nb), indexing="ij"), -1).reshape(-1, 4)
        self._ijkl = np.ascontiguousarray(np.asarray(ijkl).T, dtype=np.int32)  # (4, M)
        self._walkers = None

    @property
    def _aux_configs(self):
        return None if self._walkers is None else [w.configs for w in self._walkers]

    def _pair_ratios(self, configs, wf, there_a, there_b):
        """R (nconf, nea, neb): electron a at ``there_a`` and electron b at ``there_b``; 0 where a and b are one electron."""
        ea, eb = self._electrons
        R = np.zeros((configs.configs.shape[0], len(ea), len(eb)), dtype=wf.dtype)
        for ia, a in enumerate(ea):
            partners = eb != a
            first, saved = wf.testvalue(a, there_a)
            wf.updateinternals(a, there_a, configs, saved_values=saved)
            R[:, ia, partners] = first[:, None] * wf.testvalue_many(eb[partners], there_b)
            wf.updateinternals(a, configs.electron(a), configs)  # and back: the state ends where it started
        return R

    def _sample(self, configs, wf):
        ev, nconf = self.orbitals, configs.configs.shape[0]
        if self._walkers is None:
            naux = nconf if self._naux is None else self._naux
            self._walkers = [AuxiliaryWalkers(ev, 0), AuxiliaryWalkers(ev, 1)]
            for w in self._walkers:  # the reference warms both walks up on the spin-0 orbitals with the default step (tbdm.py:133-135)
                w.start(naux, int(np.sum(self._mol.nelec)))
                w.advance(0, self._warmup, 0.5)
        kept, pick = [], []
        for s, w in enumerate(self._walkers):
            kept.append(w.advance(s, self._nsweeps, self._tstep, keep=self._nsweeps)[1])
            pick.append(np.random.randint(0, len(w.x), size=(self._nsweeps, nconf)).astype(np.int32))  # after the walk (tbdm.py:156-161)
        x = ev.true_positions(configs)
        for s in (0, 1):
            ev.points(s, s, x[:, self._electrons[s]])
        cplx = False
        for sw in range(self._nsweeps):
            there = [ev.container(kept[s][sw][pick[s][sw]]).electron(0) for s in (0, 1)]
            R = np.ascontiguousarray(self._pair_ratios(configs, wf, *there))
            rc = np.iscomplexobj(R)
            cplx = rc or ev.dev.cplx
            ev.dev.call("pqa_tbdm_accumulate", sw, nconf, R.shape[1], R.shape[2], _ffi.ptr(pick[0][sw]), _ffi.ptr(pick[1][sw]), _ffi.ptr(R),
                        int(rc), _ffi.ptr(self._ijkl), self._ijkl.shape[1], int(sw == 0))
        return cplx

    def _result(self, configs, wf, mean):
        cplx, nconf, scale = self._sample(configs, wf), configs.configs.shape[0], 1.0 / self._nsweeps
        na, nb = self.orbitals.nmo()
        return {"value": self.orbitals.fetch(0, nconf, (self._ijkl.shape[1],), scale, mean, cplx),
                "norm_a": self.orbitals.fetch(1, nconf, (na,), scale, mean), "norm_b": self.orbitals.fetch(2, nconf, (nb,), scale, mean)}

    def __call__(self, configs, wf):
        return self._result(configs, wf, False)

    def avg(self, configs, wf):
        """Mean over the configurations, reduced on the device."""
        return self._result(configs, wf, True)

    def keys(self):
        return {"value", "norm_a", "norm_b"}

    def shapes(self):
        nmo = self.orbitals.nmo()
        return {"value": (self._ijkl.shape[1],), "norm_a": (nmo[self._spin_sector[0]],), "norm_b": (nmo[self._spin_sector[1]],)}


def normalize_tbdm(tbdm, norm_a, norm_b):
    """tbdm_ijkl / sqrt(norm_a_i norm_a_j norm_b_k norm_b_l): ratio of averages of Eq. (10), PySCF index convention
    (tbdm.py:286-290)."""
    ra, rb = np.sqrt(norm_a), np.sqrt(norm_b)
    return tbdm / (ra[:, None, None, None] * ra[None, :, None, None] * rb[None, None, :, None] * rb[None, None, None, :])
