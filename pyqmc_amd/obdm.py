"""One-body density matrix accumulator — SURVEY.md §8(f3), counterpart of ``pyqmc/observables/obdm.py``.

rho[i][j] = <c+_i c_j> is sampled by moving one electron of Psi to an auxiliary position r' drawn from
f(r) = sum_i |phi_i(r)|^2 (obdm.py:26-49).  Everything numerical runs in the HIP library:

* the basis orbitals phi_i at the auxiliary walkers and at the electrons come from ``OrbitalEvaluator`` — a device handle
  holding only ``orb_coeff`` (the fused AO evaluation + AO x MO MFMA kernel ``k_orb`` behind ``pqa_eval_mo``);
* Psi(R')/Psi(R) for every listed electron comes from ``wf.testvalue_many`` (``k_testvalue_many``).

What is left here is the reference's host logic: the Metropolis walk of the auxiliary walkers (``sample_onebody``
obdm.py:215-250), the random assignment of auxiliary walkers to configurations and the final contraction
(``OBDMAccumulator.__call__`` :139-193).  Random numbers are drawn from ``numpy.random`` in the reference's order, so a
seeded run reproduces the reference draw for draw.
"""

import numpy as np

from . import pbc as _pbc
from .systems import initial_guess
from .wf import DeviceWF


class _OneElectronView:
    """``mol`` with one electron per spin: lets a device handle carry ``norb`` orbitals without a determinant to fill."""

    def __init__(self, mol):
        object.__setattr__(self, "_mol", mol)
        object.__setattr__(self, "nelec", (1, 1))

    def __getattr__(self, name):
        return getattr(object.__getattribute__(self, "_mol"), name)


class OrbitalEvaluator:
    """``orb_coeff`` as an orbital evaluator on the device: the role ``MoleculeOrbitalEvaluator(mol, [C_up, C_dn])`` /
    ``PBCOrbitalEvaluatorKpoints(mol, [C_up, C_dn], kpts)`` play in obdm.py:79-91 and tbdm.py:87-104.  ``orb_coeff`` is
    one (nao, norb) matrix used for both spins or a pair of them.  Periodic: per spin a list ``[k]`` of (nao_prim, norb_k)
    blocks at the k-points ``kpts`` that fold onto the supercell; the orbitals are concatenated over k."""

    def __init__(self, mol, orb_coeff, kpts=None, eval_gto_precision=None, device=0):
        twist = None
        if kpts is None:
            if hasattr(mol, "a"):
                raise ValueError("kpts is required if the system is periodic")
            single = isinstance(orb_coeff, np.ndarray) and orb_coeff.ndim == 2
            pair = [np.asarray(orb_coeff)] * 2 if single else [np.asarray(orb_coeff[0]), np.asarray(orb_coeff[1])]
        else:
            if not hasattr(mol, "original_cell"):
                mol = _pbc.get_supercell(mol, np.eye(3))
            kpts = np.asarray(kpts, dtype=float).reshape(-1, 3)
            single = isinstance(orb_coeff[0], np.ndarray) and orb_coeff[0].ndim == 2  # [k] blocks shared by both spins
            per_spin = [list(orb_coeff), list(orb_coeff)] if single else [list(orb_coeff[0]), list(orb_coeff[1])]
            pair = _pbc.fold_mo_coeff(mol, kpts, per_spin)
            twist = _pbc.common_twist(mol, kpts)
        self._nmo = [int(c.shape[-1]) for c in pair]
        self.norb = self._nmo[0]
        self.mol = mol
        dets = [(1.0, [[self._nmo[0] - 1], [self._nmo[1] - 1]])]  # nmo_s = highest occupied index + 1
        kw = {} if eval_gto_precision is None else {"eval_gto_precision": eval_gto_precision}
        self.dev = DeviceWF(_OneElectronView(mol), mo_coeff=pair, determinants=dets, device=device, twist_k=twist, **kw)
        self.mo_dtype = complex if self.dev.cplx else float

    def nmo(self):
        return list(self._nmo)

    def mos(self, points, spin=0):
        """(npts, norb_spin) orbital values at ``points`` (npts, 3) (any position: periodic handles fold internally)."""
        return self.dev.eval_mo(int(spin), np.asarray(points, dtype=float).reshape(-1, 3), 1)[0]


def sample_onebody(configs, orbitals, nsamples=1, tstep=0.5, spin=0):
    """Metropolis samples of f(r) = sum_i |phi_i(r)|^2 (orbitals of ``spin``) for the one-electron walkers ``configs``
    (n,1,3) (obdm.py:215-250).  Returns (accept (nsamples,n), list of configs, list of orbital values (n,norb))."""
    n = configs.configs.shape[0]
    borb = orbitals.mos(configs.configs, spin)
    fsum = (np.abs(borb) ** 2).sum(axis=1)
    allaccept, allconfigs, allorbs = np.zeros((nsamples, n)), [], []
    for s in range(nsamples):
        shift = np.sqrt(tstep) * np.random.randn(*configs.configs.shape)
        newconfigs = configs.make_irreducible(0, (configs.configs + shift)[:, 0])
        borbnew = orbitals.mos(newconfigs.configs, spin)
        fsumnew = (np.abs(borbnew) ** 2).sum(axis=1)
        accept = fsumnew / fsum > np.random.rand(n)
        configs.move(0, newconfigs, accept)
        borb[accept] = borbnew[accept]
        fsum[accept] = fsumnew[accept]
        allconfigs.append(configs.copy())
        allaccept[s] = accept
        allorbs.append(borb.copy())
    return allaccept, allconfigs, allorbs


class OBDMAccumulator:
    """``rho[i][j] = <c+_i c_j>`` in the basis ``orb_coeff`` (obdm.py:26-213): keys ``value`` (norb,norb), ``norm`` (norb,).

    ``spin`` 0/1 restricts to the up/down electrons, ``electrons`` to an explicit list; ``naux`` auxiliary walkers
    (default: one per configuration), ``nsweeps`` auxiliary moves per evaluation, ``warmup`` moves before the first."""

    def __init__(self, mol, orb_coeff, nsweeps=5, tstep=0.50, warmup=10000, naux=None, spin=None, electrons=None, kpts=None,
                 eval_gto_precision=None, device=0, orbitals=None):
        if spin is not None:
            if spin == 0:
                self._electrons = np.arange(0, mol.nelec[0])
            elif spin == 1:
                self._electrons = np.arange(mol.nelec[0], np.sum(mol.nelec))
            else:
                raise ValueError("Spin not equal to 0 or 1")
        elif electrons is not None:
            self._electrons = np.asarray(electrons)
        else:
            self._electrons = np.arange(0, np.sum(mol.nelec))
        # `orbitals`: an object with mos(points) -> (npts, norb), .norb, .mol, .mo_dtype replacing the device evaluator
        # (the CPU tests inject the oracle's; the product path always builds the device one and fails without the library)
        self.orbitals = orbitals if orbitals is not None else OrbitalEvaluator(mol, orb_coeff, kpts=kpts, eval_gto_precision=eval_gto_precision, device=device)
        self._mol = self.orbitals.mol
        self.dtype = self.orbitals.mo_dtype
        self._tstep = tstep
        self.nelec = len(self._electrons)
        self._nsweeps = self._nstep = nsweeps
        self._warmup = warmup
        self._naux = naux
        self._warmed_up = False
        self.norb = self.orbitals.norb

    def warm_up(self, naux):
        self._extra_config = initial_guess(self._mol, int(naux / self.nelec) + 1, rng=np.random)
        self._extra_config.reshape((-1, 1, 3))
        self._extra_config.resample(np.arange(naux))
        _, extra_configs, _ = sample_onebody(self._extra_config, self.orbitals, nsamples=self._warmup, tstep=self._tstep)
        self._extra_config = extra_configs[-1]

    def __call__(self, configs, wf):
        nconf = configs.configs.shape[0]
        if not self._warmed_up:
            self.warm_up(nconf if self._naux is None else self._naux)
            self._warmed_up = True
        results = {"value": np.zeros((nconf, self.norb, self.norb), dtype=self.dtype), "norm": np.zeros((nconf, self.norb))}
        naux = self._extra_config.configs.shape[0]
        auxassignments = np.random.randint(0, naux, size=(self._nsweeps, nconf))
        _, extra_configs, borb_aux = sample_onebody(self._extra_config, self.orbitals, nsamples=self._nsweeps, tstep=self._tstep)
        self._extra_config = extra_configs[-1]
        for conf, assign in zip(extra_configs, auxassignments):
            conf.resample(assign)
        borb_aux = np.asarray([orb[assign, ...] for orb, assign in zip(borb_aux, auxassignments)])
        borb_configs = self.evaluate_orbitals(configs.select_electrons(self._electrons)).reshape(nconf, self.nelec, -1)
        bauxsquared = np.abs(borb_aux) ** 2
        fsum = np.sum(bauxsquared, axis=-1, keepdims=True) / self.norb
        norm = bauxsquared / fsum
        baux_f = borb_aux / fsum
        for sweep in range(self._nsweeps):
            wfratio = wf.testvalue_many(self._electrons, extra_configs[sweep].electron(0))
            results["value"] += np.einsum("ie,ij,iek->ijk", wfratio.conj(), baux_f[sweep], borb_configs.conj(), optimize=True)
            results["norm"] += norm[sweep]
        results["value"] /= self._nstep
        results["norm"] = results["norm"] / self._nstep
        return results

    def avg(self, configs, wf):
        return {k: np.mean(it, axis=0) for k, it in self(configs, wf).items()}

    def evaluate_orbitals(self, configs):
        return self.orbitals.mos(configs.configs)

    def keys(self):
        return set(["value", "norm"])

    def shapes(self):
        return {"value": (self.norb, self.norb), "norm": (self.norb,)}


def normalize_obdm(obdm, norm):
    return obdm / (norm[np.newaxis, :] * norm[:, np.newaxis]) ** 0.5
