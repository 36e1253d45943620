"""``EnergyAccumulator`` with the interface of ``pyqmc/observables/accumulators.py:45-95``
(``__call__(configs, wf) -> dict``, ``avg``, ``keys``, ``shapes``, ``has_nonlocal_moves``),
evaluated by the fused HIP energy path (kinetic ``energy.py:57-65``, open-boundary Coulomb
``energy.py:28-54``, ECP ``eval_ecp.py:21-146``) on the walkers resident behind ``wf``.

The accumulator needs a wave function whose factors share one device handle
(``pyqmc_amd.generate_wf``); there is deliberately no host-side fallback.
"""

import numpy as np

KEYS = ("ke", "ee", "ei", "ecp", "grad2", "total")
NAIP = (6, 12, 18, 26, 32, 50)  # the quadrature grids of eval_ecp.py:278-336


class EnergyAccumulator:
    def __init__(self, mol, threshold=10, naip=None, use_old_ecp=True, seed=None, check_configs=True, **kwargs):
        """``kwargs``: ``ewald_gmax`` / ``nlatvec`` of the periodic Coulomb sum (accumulators.py:48-53, ewald.py:95).
        ``naip``: number of quadrature points of the ECP integrator for every ECP atom (accumulators.py:48-51, eval_ecp.py:228-252);
        None (default) is the reference's per-atom choice, 6 or 12 points by channel count (eval_ecp.py:239-240).
        ``use_old_ecp=False``: the batched ECP integrator (jax_ecp.ECPAccumulator, accumulators.py:57-58; ``pyqmc_amd.ECPAccumulator``);
        ``threshold`` plays no part there, ``rot`` / ``unif`` of ``__call__`` then have that integrator's layout.
        ``seed``: key of the device's ECP rotation / mask streams; None (default) draws a fresh key from ``numpy.random``
        at every evaluation, so ``np.random.seed`` controls the run as it does in the reference (eval_ecp.py:255-275, :135-146)
        and accumulators on different ranks do not replay one another's rotations; an integer makes the sequence explicit."""
        self.mol = mol
        self._ewald_kws = kwargs
        if kwargs and not hasattr(mol, "a"):
            raise TypeError(f"unexpected arguments {sorted(kwargs)} for an open-boundary system")
        self.threshold = threshold
        if naip is not None and naip not in NAIP:  # eval_ecp.get_rot refuses anything else (eval_ecp.py:266-267)
            raise ValueError(f"Possible AIPs are one of {NAIP}")
        self.naip = naip
        self.use_old_ecp = use_old_ecp
        if not use_old_ecp:  # accumulators.py:57-58
            from .ecp_batched import ECPAccumulator

            self.ecp = ECPAccumulator(mol, naip=naip)
        self.seed = None if seed is None else int(seed)
        self._calls = 0
        self.check_configs = check_configs

    @staticmethod
    def _device(wf):
        dev = wf.fused_device() if hasattr(wf, "fused_device") else getattr(wf, "_dev", None)
        if dev is None:
            raise TypeError("pyqmc_amd.EnergyAccumulator needs a pyqmc_amd wave function living on one device handle")
        return dev

    def bind(self, dev):
        """Make the handle's energy pass the one this accumulator describes (quadrature rule, Ewald tables)."""
        if self.use_old_ecp:
            dev.set_ecp_batched(None)
            dev.set_ecp_naip(self.naip)
        else:
            self.ecp.bind(dev)
        if dev.pbc:
            dev.set_ewald(**self._ewald_kws)

    def __call__(self, configs, wf, rot=None, unif=None):
        dev = self._device(wf)
        if getattr(dev, "twisted", False):  # twisted handles hold unfolded coordinates (include/pyqmc_amd.h)
            same = np.allclose(dev.configs(), configs.configs + configs.wrap @ configs.lvecs, rtol=0, atol=1e-9)
        else:
            same = np.array_equal(dev.configs(), configs.configs)
        if self.check_configs and not same:
            raise ValueError("walkers on the device differ from `configs`: call wf.recompute(configs) "
                             "(or keep wf.updateinternals in step with configs.move) first")
        self.bind(dev)
        self._calls += 1
        if rot is not None and unif is not None:
            key = 0  # every draw is replayed: the device streams are not used
        else:
            key = int(np.random.randint(0, 2**31 - 1)) if self.seed is None else self.seed + self._calls
        out = dev.energy(self.threshold, rot=rot, unif=unif, seed=key)
        if np.iscomplexobj(out):  # complex orbitals: ecp and total are complex (eval_ecp.py:89), the rest real (energy.py:62-64)
            return {k: (out[i] if k in ("ecp", "total") else out[i].real.copy()) for i, k in enumerate(KEYS)}
        return {k: out[i] for i, k in enumerate(KEYS)}

    def avg(self, configs, wf):
        return {k: np.mean(v, axis=0) for k, v in self(configs, wf).items()}

    def nonlocal_tmoves(self, configs, wf, e, tau, rot=None, unif=None):
        """``EnergyAccumulator.nonlocal_tmoves`` (accumulators.py:80-86) -> ``eval_ecp.compute_tmoves``
        (eval_ecp.py:43-80): dict with ``ratio`` (W,P), ``weight`` (W,P) and ``configs`` (an electron object with
        (W,P,3) candidate positions) over all ECP atoms' quadrature points.  ``rot`` (necp,3,3) / ``unif`` (necp,W)
        replay the reference's draws; by default they are drawn from ``numpy.random`` like the reference does."""
        from . import _ffi

        if not self.use_old_ecp:  # accumulators.py:84-86
            return self.ecp.nonlocal_tmoves(configs, wf, e, tau, rot=rot, unif=unif)
        dev = self._device(wf)
        W, P = dev.W, dev.call_int("pqa_tmove_npoints")
        if P == 0:  # no ECP atom: empty candidate lists (the callers index all three keys, dmc.py:96-101)
            return {"ratio": np.ones((W, 0)), "weight": np.zeros((W, 0)), "configs": configs.make_irreducible(e, np.zeros((W, 0, 3)))}
        necp = dev.necp
        if unif is None:
            unif = np.random.random(size=(necp, W))
        if rot is None:
            q = np.random.normal(size=(necp, 4))
            q /= np.linalg.norm(q, axis=1, keepdims=True)
            w_, x, y, z = q.T
            rot = np.stack([np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w_), 2 * (x * z + y * w_)], -1),
                            np.stack([2 * (x * y + z * w_), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w_)], -1),
                            np.stack([2 * (x * z - y * w_), 2 * (y * z + x * w_), 1 - 2 * (x * x + y * y)], -1)], -2)
        rot, unif = _ffi.f64(rot), _ffi.f64(unif)
        ratio, weight, pos = np.empty((W, P)), np.empty((W, P)), np.empty((W, P, 3))
        if getattr(dev, "cplx", False):
            # complex determinants: candidate positions and weights from the device, the (complex) ratios through the protocol's
            # testvalue at those positions; candidates of atoms that failed the ECP mask keep ratio 1 like the reference's table
            dev.call("pqa_tmoves", int(e), float(tau), float(self.threshold), _ffi.ptr(rot), _ffi.ptr(unif), None, _ffi.ptr(weight), _ffi.ptr(pos))
            epos = configs.make_irreducible(e, pos)
            live = weight != 0.0
            ratio = np.ones((W, P), dtype=complex)
            if live.any():
                ratio[live] = np.asarray(wf.testvalue(e, epos)[0])[live]
            return {"ratio": ratio, "weight": weight, "configs": epos}
        dev.call("pqa_tmoves", int(e), float(tau), float(self.threshold), _ffi.ptr(rot), _ffi.ptr(unif), _ffi.ptr(ratio),
                 _ffi.ptr(weight), _ffi.ptr(pos))
        return {"ratio": ratio, "weight": weight, "configs": configs.make_irreducible(e, pos)}

    def has_nonlocal_moves(self):
        return self.mol._ecp != {}

    def keys(self):
        return set(KEYS)

    def shapes(self):
        return {k: () for k in KEYS}
