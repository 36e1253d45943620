"""One-body density matrix accumulator — SURVEY.md §8(f3); public interface of ``pyqmc/observables/obdm.py``.

``rho[i][j] = <c+_i c_j>`` in the orbital basis ``orb_coeff`` is sampled by moving one electron of Psi to an auxiliary
position r' distributed as ``f(r) = sum_i |phi_i(r)|^2`` (obdm.py:26-49).  The work is done by the HIP library on a
device handle that holds ``orb_coeff`` as its orbitals (``OrbitalEvaluator``):

* ``pqa_dm_walk``: the Metropolis walk of the auxiliary walkers (obdm.py:215-250), resident on the device — one orbital
  launch (``k_orb``) and one accept kernel per sample; the last ``nsweeps`` samples stay on the device;
* ``pqa_dm_points``: the basis orbitals at the configurations' electrons;
* ``pqa_obdm_accumulate``: per sweep, the estimator's contraction (obdm.py:170-190) with the ratios Psi(R')/Psi(R) of
  ``wf.testvalue_many`` (``k_testvalue_many`` on the wave function's own handle);
* ``pqa_dm_fetch``: the per-configuration result, or its mean over configurations reduced on the device (``avg``).

The host only draws random numbers — from ``numpy.random`` in the reference's order, so that a seeded run reproduces the
reference draw for draw — and moves the small per-sweep arrays (positions, assignments, ratios).
"""

import numpy as np

from . import _ffi
from . import pbc as _pbc
from .configs import OpenConfigs, PeriodicConfigs
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
        self.lattice = np.asarray(mol.lattice_vectors(), dtype=float) if hasattr(mol, "a") else None

    def nmo(self):
        return list(self._nmo)

    def true_positions(self, obj):
        """Coordinates the device is handed: a container's positions plus, for a periodic one, ``wrap @ lattice`` — the
        orbital kernel folds every point itself and derives a twisted cell's wrap phase exp(i k . wrap . L)
        (orbitals.py:201-213) from the fold, so electrons that left the cell carry the right Bloch phase."""
        x = np.asarray(obj.configs, dtype=float)
        if self.lattice is not None and getattr(obj, "wrap", None) is not None:
            x = x + np.asarray(obj.wrap, dtype=float) @ self.lattice
        return x

    def container(self, x):
        """One-electron configurations (n,1,3) for true positions ``x`` (n,3): folded into the cell, with wrap counters."""
        x = np.asarray(x, dtype=float).reshape(-1, 1, 3)
        return OpenConfigs(x.copy()) if self.lattice is None else PeriodicConfigs(x, self.lattice)

    def mos(self, points, spin=0):
        """(npts, norb_spin) orbital values at ``points`` (npts, 3) (any position: periodic handles fold internally)."""
        return self.dev.eval_mo(int(spin), np.asarray(points, dtype=float).reshape(-1, 3), 1)[0]

    # ---- device-resident pieces of the estimators --------------------------------------------------------------
    def walk(self, slot, spin, x, gauss, unif, tstep, nkeep):
        """``pqa_dm_walk`` with replay tapes: advances the true positions ``x`` (n,3) in place; returns the decisions
        (nsamples,n) and the positions of the last ``nkeep`` samples (nkeep,n,3)."""
        nsamples, n = unif.shape
        keep, acc = np.empty((nkeep, n, 3)), np.empty((nsamples, n))
        self.dev.call("pqa_dm_walk", int(slot), int(spin), n, nsamples, float(tstep), _ffi.ptr(x), _ffi.ptr(gauss), _ffi.ptr(unif), 0,
                      int(nkeep), _ffi.ptr(keep), _ffi.ptr(acc))
        return acc, keep

    def points(self, slot, spin, x):
        x = _ffi.f64(x).reshape(-1, 3)
        self.dev.call("pqa_dm_points", int(slot), int(spin), _ffi.ptr(x), len(x))

    def fetch(self, which, nconf, shape, scale, mean, cplx=False):
        """``pqa_dm_fetch``: accumulator ``which`` (0 value, 1 / 2 norms) times ``scale``, per configuration
        (nconf, *shape) or averaged over the configurations on the device (*shape)."""
        ncol = int(np.prod(shape)) * (2 if cplx else 1)
        out = np.empty(((1 if mean else nconf), ncol))
        self.dev.call("pqa_dm_fetch", int(which), ncol, float(scale), int(mean), _ffi.ptr(out))
        out = out.view(complex) if cplx else out
        return out.reshape(shape) if mean else out.reshape((nconf,) + tuple(shape))


def draw_walk_tapes(n, nsamples):
    """Standard normals (nsamples,n,3) and uniforms (nsamples,n) in the order the reference's walk consumes
    ``numpy.random`` (per sample: the displacements of all walkers, then their acceptance numbers; obdm.py:232-240)."""
    gauss, unif = np.empty((nsamples, n, 3)), np.empty((nsamples, n))
    for s in range(nsamples):
        gauss[s] = np.random.randn(n, 3)
        unif[s] = np.random.rand(n)
    return gauss, unif


class AuxiliaryWalkers:
    """``n`` one-electron walkers distributed as the orbital density of one spin, resident in a slot of the evaluator's
    device handle.  ``start`` places them like the reference (``initial_guess`` electrons re-read one by one,
    obdm.py:121-124); ``advance`` runs the walk and keeps the last ``keep`` samples on the device."""

    def __init__(self, orbitals, slot):
        self.orbitals, self.slot, self.x = orbitals, slot, None

    def start(self, naux, electrons_per_config):
        seed = initial_guess(self.orbitals.mol, int(naux / electrons_per_config) + 1, rng=np.random)
        self.x = np.ascontiguousarray(self.orbitals.true_positions(seed).reshape(-1, 3)[:naux])

    def advance(self, spin, nsamples, tstep, keep=0):
        gauss, unif = draw_walk_tapes(len(self.x), nsamples)
        return self.orbitals.walk(self.slot, spin, self.x, gauss, unif, tstep, keep)

    @property
    def configs(self):
        return self.orbitals.container(self.x)


def sample_onebody(configs, orbitals, nsamples=1, tstep=0.5, spin=0):
    """The reference's free function (obdm.py:215-250) on the device walk: advances the one-electron ``configs`` (n,1,3)
    and returns (decisions (nsamples,n), list of configurations, list of orbital values (n,norb)) per sample."""
    w = AuxiliaryWalkers(orbitals, 0)
    w.x = np.ascontiguousarray(orbitals.true_positions(configs).reshape(-1, 3))
    acc, kept = w.advance(spin, nsamples, tstep, keep=nsamples)
    snaps = [orbitals.container(k) for k in kept]
    if nsamples:
        last = snaps[-1]
        configs.configs[...] = last.configs
        if getattr(configs, "wrap", None) is not None:
            configs.wrap[...] = last.wrap
    return acc, snaps, [orbitals.mos(k, spin) for k in kept]


class OBDMAccumulator:
    """Keys ``value`` (norb,norb), ``norm`` (norb,) per configuration (obdm.py:26-213).

    ``spin`` 0/1 restricts the moved electrons to the up/down ones, ``electrons`` to an explicit list; ``naux`` auxiliary
    walkers (default: one per configuration), ``nsweeps`` auxiliary samples per evaluation, ``warmup`` samples before the
    first."""

    def __init__(self, mol, orb_coeff, nsweeps=5, tstep=0.50, warmup=10000, naux=None, spin=None, electrons=None, kpts=None,
                 eval_gto_precision=None, device=0):
        nup, ntot = mol.nelec[0], int(np.sum(mol.nelec))
        if spin is not None:
            if spin not in (0, 1):
                raise ValueError("Spin not equal to 0 or 1")
            self._electrons = np.arange(0, nup) if spin == 0 else np.arange(nup, ntot)
        else:
            self._electrons = np.arange(ntot) if electrons is None else np.asarray(electrons)
        self.orbitals = OrbitalEvaluator(mol, orb_coeff, kpts=kpts, eval_gto_precision=eval_gto_precision, device=device)
        self._mol, self.dtype, self.norb = self.orbitals.mol, self.orbitals.mo_dtype, self.orbitals.norb
        self.nelec = len(self._electrons)
        self._tstep, self._nsweeps, self._warmup, self._naux = tstep, nsweeps, warmup, naux
        self._walkers = None

    @property
    def _extra_config(self):
        return None if self._walkers is None else self._walkers.configs

    def _sample(self, configs, wf):
        """Runs one evaluation on the device; returns whether the accumulated value is complex."""
        ev, nconf = self.orbitals, configs.configs.shape[0]
        if self._walkers is None:
            self._walkers = AuxiliaryWalkers(ev, 0)
            self._walkers.start(nconf if self._naux is None else self._naux, self.nelec)
            self._walkers.advance(0, self._warmup, self._tstep)
        naux = len(self._walkers.x)
        pick = np.random.randint(0, naux, size=(self._nsweeps, nconf)).astype(np.int32)  # drawn before the walk (obdm.py:150)
        _, kept = self._walkers.advance(0, self._nsweeps, self._tstep, keep=self._nsweeps)
        ev.points(0, 0, ev.true_positions(configs)[:, self._electrons])
        cplx = False
        for s in range(self._nsweeps):
            there = ev.container(kept[s][pick[s]]).electron(0)
            ratio = np.ascontiguousarray(wf.testvalue_many(self._electrons, there))
            rc = np.iscomplexobj(ratio)
            cplx = rc or ev.dev.cplx
            ev.dev.call("pqa_obdm_accumulate", 0, s, nconf, self.nelec, _ffi.ptr(pick[s]), _ffi.ptr(ratio), int(rc), int(s == 0))
        # the reference resamples its last sample in place and walks on from THAT set (one walker per configuration,
        # obdm.py:160-163); kept so that a seeded run stays draw-for-draw comparable
        self._walkers.x = np.ascontiguousarray(kept[-1][pick[-1]])
        return cplx

    def _result(self, configs, wf, mean):
        cplx, nconf, scale = self._sample(configs, wf), configs.configs.shape[0], 1.0 / self._nsweeps
        return {"value": self.orbitals.fetch(0, nconf, (self.norb, self.norb), scale, mean, cplx),
                "norm": self.orbitals.fetch(1, nconf, (self.norb,), scale, mean)}

    def __call__(self, configs, wf):
        return self._result(configs, wf, False)

    def avg(self, configs, wf):
        """Mean over the configurations, reduced on the device (obdm.py:195-197)."""
        return self._result(configs, wf, True)

    def evaluate_orbitals(self, configs):
        return self.orbitals.mos(self.orbitals.true_positions(configs))

    def keys(self):
        return {"value", "norm"}

    def shapes(self):
        return {"value": (self.norb, self.norb), "norm": (self.norb,)}


def normalize_obdm(obdm, norm):
    """rho_ij / sqrt(norm_i norm_j) (obdm.py:252-253)."""
    return obdm / np.sqrt(np.outer(norm, norm))
