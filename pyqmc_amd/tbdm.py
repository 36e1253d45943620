"""Two-body density matrix accumulator — SURVEY.md §8(f3), counterpart of ``pyqmc/observables/tbdm.py``.

One spin sector ``tbdm[s1,s2][i,j,k,l] = <c+_{s1,i} c+_{s2,k} c_{s2,l} c_{s1,j}>`` (PySCF's index convention), sampled
by moving an electron pair (a, b) to auxiliary positions drawn from the orbital densities (Eq. 10 of
DOI:10.1063/1.4793531).  Device work: the basis orbitals through ``obdm.OrbitalEvaluator`` (``k_orb``), the first
electron's move through ``testvalue`` / ``updateinternals`` (Sherman-Morrison on the device), the second electron's
ratios for all partners at once through ``testvalue_many`` (``k_testvalue_many``).  Host logic, draw order and output
keys are the reference's (``tbdm.py:63-283``).
"""

import numpy as np

from . import obdm
from .systems import initial_guess


class TBDMAccumulator:
    """Keys ``value`` (M,), ``norm_a`` (norb_s1,), ``norm_b`` (norb_s2,) for the M index tuples ``ijkl`` (default: all).

    ``orb_coeff`` (2, nao, norb): basis of the 2-RDM per spin; ``spin`` = (s1, s2) sector."""

    def __init__(self, mol, orb_coeff, spin, nsweeps=4, tstep=0.50, warmup=200, naux=None, ijkl=None, kpts=None,
                 eval_gto_precision=None, device=0, orbitals=None):
        self._tstep, self._nsweeps, self._spin, self._naux, self._warmup = tstep, nsweeps, spin, naux, warmup
        # `orbitals`: see obdm.OBDMAccumulator (CPU tests inject the oracle's evaluator)
        self.orbitals = orbitals if orbitals is not None else obdm.OrbitalEvaluator(mol, orb_coeff, kpts=kpts, eval_gto_precision=eval_gto_precision, device=device)
        self._mol = self.orbitals.mol
        norb_up, norb_down = self.orbitals.nmo()
        self.dtype = self.orbitals.mo_dtype
        self._spin_sector = spin
        nelec = self._mol.nelec
        self._electrons = [np.arange(spin[s] * nelec[0], nelec[0] + spin[s] * nelec[1]) for s in (0, 1)]
        if ijkl is None:  # the full 2-RDM sector
            ijkl = [[i, j, k, l] for i in range(norb_up) for j in range(norb_up) for k in range(norb_down) for l in range(norb_down)]
        self._ijkl = np.array(ijkl).T
        self._warmed_up = False

    def warm_up(self, naux):
        nwalkers = int(naux / sum(self._mol.nelec)) + 1
        self._aux_configs = []
        for spin in (0, 1):
            self._aux_configs.append(initial_guess(self._mol, nwalkers, rng=np.random))
            self._aux_configs[spin].reshape((-1, 1, 3))
            self._aux_configs[spin].resample(np.arange(naux))
            # (the reference warms both walks up on the spin-0 orbitals, tbdm.py:133-135)
            _, cfgs, _ = obdm.sample_onebody(self._aux_configs[spin], self.orbitals, nsamples=self._warmup, spin=0)
            self._aux_configs[spin] = cfgs[-1]

    def get_configurations(self, nconf):
        """One auxiliary configuration per walker and sweep, per spin (tbdm.py:139-186)."""
        configs, assignments, orbs, acceptance = [], [], [], []
        for spin in (0, 1):
            naux = self._aux_configs[spin].configs.shape[0]
            accept, tmp_config, tmp_orbs = obdm.sample_onebody(self._aux_configs[spin], self.orbitals, nsamples=self._nsweeps,
                                                               tstep=self._tstep, spin=spin)
            assignments.append(np.random.randint(0, naux, size=(self._nsweeps, nconf)))
            self._aux_configs[spin] = tmp_config[-1].copy()
            acceptance.append(accept)
            for conf, assign in zip(tmp_config, assignments[-1]):
                conf.resample(assign)
            configs.append(tmp_config)
            orbs.append([orb[assign, ...] for orb, assign in zip(tmp_orbs, assignments[-1])])
        return {"acceptance": acceptance, "orbs": orbs, "configs": configs, "assignments": assignments}

    def __call__(self, configs, wf):
        nconf, nelec = configs.configs.shape[:2]
        if not self._warmed_up:
            self.warm_up(nconf if self._naux is None else self._naux)
            self._warmed_up = True
        aux = self.get_configurations(nconf)
        orb_configs = []
        for s in (0, 1):
            es = self._electrons[s]
            o = self.orbitals.mos(configs.configs[:, es].reshape(-1, 3), s).reshape(nconf, len(es), -1)
            orb_configs.append(o)
        results = {"value": np.zeros((nconf, self._ijkl.shape[1]), dtype=self.dtype),
                   "norm_a": np.zeros((nconf, orb_configs[0].shape[-1])), "norm_b": np.zeros((nconf, orb_configs[1].shape[-1]))}
        orb_configs = [orb_configs[s][:, :, self._ijkl[2 * s]] for s in (0, 1)]
        down_start = [np.min(self._electrons[s]) for s in (0, 1)]
        for sweep in range(self._nsweeps):
            fsum = [np.sum(np.abs(aux["orbs"][s][sweep]) ** 2, axis=1) for s in (0, 1)]
            norm = [np.abs(aux["orbs"][s][sweep]) ** 2 / fsum[s][:, np.newaxis] for s in (0, 1)]
            wfratio, electrons_a_ind, electrons_b_ind = [], [], []
            for ea in self._electrons[0]:
                electrons_b = self._electrons[1][self._electrons[1] != ea]  # don't move the same electron twice
                epos_a = aux["configs"][0][sweep].electron(0)
                epos_b = aux["configs"][1][sweep].electron(0)
                wfratio_a, saved_a = wf.testvalue(ea, epos_a)
                wf.updateinternals(ea, epos_a, configs, saved_values=saved_a)
                wfratio_b = wf.testvalue_many(electrons_b, epos_b)
                wf.updateinternals(ea, configs.electron(ea), configs)  # back (the orbital row is re-evaluated on the device)
                wfratio.append(wfratio_a[:, np.newaxis] * wfratio_b)
                electrons_a_ind.extend([ea - down_start[0]] * len(electrons_b))
                electrons_b_ind.extend(electrons_b - down_start[1])
            wfratio = np.concatenate(wfratio, axis=1)
            phi_j_r1p = aux["orbs"][0][sweep][..., self._ijkl[1]]
            phi_l_r2p = aux["orbs"][1][sweep][..., self._ijkl[3]]
            rho1rho2 = 1.0 / (fsum[0] * fsum[1])
            # n walker, i electron pair, o index tuple:  phi_i(r1) phi_k(r2) phi_j*(r1') phi_l*(r2') / rho(r1') rho(r2')
            orbratio = np.einsum("nio,nio,no,no,n->nio", orb_configs[0][:, electrons_a_ind, :], orb_configs[1][:, electrons_b_ind, :],
                                 phi_j_r1p.conj(), phi_l_r2p.conj(), rho1rho2)
            results["value"] += np.einsum("in,inj->ij", wfratio, orbratio)
            results["norm_a"] += norm[0]
            results["norm_b"] += norm[1]
        results["value"] /= self._nsweeps
        results["norm_a"] /= self._nsweeps
        results["norm_b"] /= self._nsweeps
        return results

    def keys(self):
        return set(["value", "norm_a", "norm_b"])

    def shapes(self):
        nmo = self.orbitals.nmo()
        d = {"value": (self._ijkl.shape[1],)}
        for e, s in zip("ab", self._spin_sector):
            d["norm_%s" % e] = (nmo[s],)
        return d

    def avg(self, configs, wf):
        return {k: np.mean(it, axis=0) for k, it in self(configs, wf).items()}


def normalize_tbdm(tbdm, norm_a, norm_b):
    """Ratio of averages of Eq. (10), PySCF index convention (tbdm.py:286-290)."""
    return tbdm / np.einsum("i,j,k,l->ijkl", norm_a, norm_a, norm_b, norm_b) ** 0.5
