"""``ECPAccumulator`` with the interface of ``pyqmc/observables/jax_ecp.py:22-142`` — the batched formulation of the ECP
integral that ``EnergyAccumulator(use_old_ecp=False)`` selects (``accumulators.py:57-58, 63-64, 84-86``): for every electron
all ECP atoms' quadrature points form one table, the ``nselect_deterministic`` points of largest ``sum_l v_l(r)^2`` are
evaluated with weight 1 and ``nselect_random`` of the others are sampled (``evaluate_vl`` :160-222, ``downselect_move_info``
:225-290).  The table, the selection, the auxiliary positions and the ratios are all evaluated on the device handle behind
``wf`` (``pqa_set_ecp_batched``, ``csrc/pqa_ecpb.hpp``); there is no host-side fallback.

Where this mirror departs from the letter of the reference, and why:
  * an integer ``naip`` means that many points at every atom with a non-local channel (the documented meaning of the old
    accumulator's argument); the reference multiplies it into an array of zeros (jax_ecp.py:55-56), which switches the
    non-local part off altogether;
  * atoms without an ECP carry no points (the reference's constructor raises on them: ``np.max`` of an empty key list, :47);
  * among equal probabilities — the points of one atom share theirs — the kept points are those a STABLE ascending sort puts
    last; numpy's default ``argsort`` (jax_ecp.py:241) is not stable, so where a tie straddles the cut the reference's own choice
    depends on the sort implementation.  Every such choice samples the same estimator.
"""

import numpy as np

NAIP = (0, 6, 12, 18, 26, 32, 50)


def default_naip(mol):
    """jax_ecp.py:43-54: by the highest non-local channel of each atom's ECP (0 -> 6, 1 -> 6, 2 -> 12, anything else 0)."""
    out = []
    for i in range(mol.natm):
        sym = mol.atom_pure_symbol(i) if mol.atom_symbol(i) not in mol._ecp else mol.atom_symbol(i)
        if sym not in mol._ecp:
            continue
        max_l = max(int(l) for l, _ in mol._ecp[sym][1])
        out.append({0: 6, 1: 6, 2: 12}.get(max_l, 0))
    return np.asarray(out, dtype=np.int32)


def _ecp_atom_count(mol):
    return sum(1 for i in range(mol.natm) if mol.atom_symbol(i) in mol._ecp or mol.atom_pure_symbol(i) in mol._ecp)


def random_rotations(n):
    """n uniformly random rotation matrices from ``numpy.random`` (the reference draws scipy's Rotation.random per atom, eval_ecp.py:263)."""
    q = np.random.normal(size=(n, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    w_, x, y, z = q.T
    return np.stack([np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w_), 2 * (x * z + y * w_)], -1),
                     np.stack([2 * (x * y + z * w_), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w_)], -1),
                     np.stack([2 * (x * z - y * w_), 2 * (y * z + x * w_), 1 - 2 * (x * x + y * y)], -1)], -2)


class ECPAccumulator:
    def __init__(self, mol, naip=None, stochastic_rotation=True, nselect_deterministic=None, nselect_random=None, check_configs=True):
        self._ecp = mol._ecp
        necp = _ecp_atom_count(mol)
        if naip is None:
            naip = default_naip(mol)
        elif np.isscalar(naip):
            naip = np.where(default_naip(mol) > 0, int(naip), 0)
        naip = np.asarray(naip, dtype=np.int32)
        if naip.shape != (necp,) or any(int(n) not in NAIP for n in naip):
            raise ValueError(f"naip: one of {NAIP} for each of the {necp} ECP atoms")
        self.naip = naip
        totaip = int(naip.sum())
        self.nselect_deterministic = int(naip.max()) if (nselect_deterministic is None and necp) else int(nselect_deterministic or 0)
        self.nselect_random = min(1, totaip - self.nselect_deterministic) if nselect_random is None else int(nselect_random)  # jax_ecp.py:66-69
        self.nselect_random = max(self.nselect_random, 0)
        self.stochastic_rotation = stochastic_rotation
        self.check_configs = check_configs

    # ---- device side
    @staticmethod
    def _device(wf):
        dev = wf.fused_device() if hasattr(wf, "fused_device") else getattr(wf, "_dev", None)
        if dev is None:
            raise TypeError("pyqmc_amd.ECPAccumulator needs a pyqmc_amd wave function living on one device handle")
        return dev

    def bind(self, dev):
        dev.set_ecp_batched(self.naip, self.nselect_deterministic, self.nselect_random)
        # the ECP row comes out of the handle's whole energy pass (kinetic, Coulomb, ECP), and on a periodic handle that pass refuses to run
        # without the Ewald tables — which only an EnergyAccumulator would have set: a standalone ECPAccumulator sets the defaults itself
        if getattr(dev, "pbc", False) and getattr(dev, "_ewald_key", None) is None:
            dev.set_ewald()

    def _rotations(self, dev, n, rot):
        if rot is not None:
            return rot
        if not self.stochastic_rotation:  # get_rot(..., stochastic=False): the grid as tabulated (eval_ecp.py:272-273)
            return np.broadcast_to(np.eye(3), (n, dev.necp, 3, 3)).copy()
        return random_rotations(n * dev.necp).reshape(n, dev.necp, 3, 3)

    def __call__(self, configs, wf, rot=None, unif=None):
        """ECP energy of every walker (jax_ecp.py:72-105).  ``rot`` (N, necp, 3, 3) / ``unif`` (N, W, nselect_random) replay the
        reference's draws (one rotation per atom with points, one uniform per random selection); drawn from ``numpy.random``
        otherwise."""
        dev = self._device(wf)
        if self.check_configs and not np.array_equal(dev.configs(), configs.configs):
            raise ValueError("walkers on the device differ from `configs`: call wf.recompute(configs) first")
        self.bind(dev)
        try:
            W, N = configs.configs.shape[:2]
            rot = self._rotations(dev, N, rot)
            if unif is None:
                unif = np.random.random((N, W, max(self.nselect_random, 1)))[:, :, :self.nselect_random]
            out = dev.energy(0.0, rot=rot, unif=np.ascontiguousarray(unif) if np.size(unif) else None, seed=0)
        finally:
            dev.set_ecp_batched(None)
        return out[3]

    def avg(self, configs, wf):
        return {"ecp": np.mean(self(configs, wf), axis=0)}

    def nonlocal_tmoves(self, configs, wf, e, tau, rot=None, unif=None):
        """jax_ecp.py:110-135: ``ratio`` (W, P), ``weight`` (W, P), ``configs`` (electron object with (W, P, 3) positions) over the
        P selected points.  ``rot`` (necp, 3, 3), ``unif`` (W, nselect_random)."""
        dev = self._device(wf)
        self.bind(dev)
        try:
            W = dev.W
            rot = self._rotations(dev, 1, None if rot is None else np.asarray(rot)[None])[0]
            if unif is None:
                unif = np.random.random((W, max(self.nselect_random, 1)))[:, :self.nselect_random]
            weight, pos = dev.ecp_batched_moves(e, tau, rot, np.ascontiguousarray(unif))
        finally:
            dev.set_ecp_batched(None)
        epos = configs.make_irreducible(e, pos)
        if pos.shape[1] == 0:
            return {"ratio": np.ones((W, 0)), "weight": weight, "configs": epos}
        ratio = np.asarray(wf.testvalue(e, epos)[0])
        return {"ratio": ratio, "weight": weight, "configs": epos}

    def has_nonlocal_moves(self):
        return self._ecp != {}

    def keys(self):
        return set(["ecp"])

    def shapes(self):
        return {"ecp": ()}
