"""``EnergyAccumulator`` with the interface of ``pyqmc/observables/accumulators.py:45-95``
(``__call__(configs, wf) -> dict``, ``avg``, ``keys``, ``shapes``, ``has_nonlocal_moves``),
evaluated by the fused HIP energy path (kinetic ``energy.py:57-65``, open-boundary Coulomb
``energy.py:28-54``, ECP ``eval_ecp.py:21-146``) on the walkers resident behind ``wf``.

The accumulator needs a wave function whose factors share one device handle
(``pyqmc_amd.generate_wf``); there is deliberately no host-side fallback.
"""

import numpy as np

KEYS = ("ke", "ee", "ei", "ecp", "grad2", "total")


class EnergyAccumulator:
    def __init__(self, mol, threshold=10, naip=None, seed=0, check_configs=True):
        self.mol = mol
        self.threshold = threshold
        if naip is not None:
            raise NotImplementedError("naip is chosen per atom as in eval_ecp.py:239-240 (6 or 12)")
        self.seed = int(seed)
        self._calls = 0
        self.check_configs = check_configs

    @staticmethod
    def _device(wf):
        dev = wf.fused_device() if hasattr(wf, "fused_device") else getattr(wf, "_dev", None)
        if dev is None:
            raise TypeError("pyqmc_amd.EnergyAccumulator needs a pyqmc_amd wave function living on one device handle")
        return dev

    def __call__(self, configs, wf, rot=None, unif=None):
        dev = self._device(wf)
        if self.check_configs and not np.array_equal(dev.configs(), configs.configs):
            raise ValueError("walkers on the device differ from `configs`: call wf.recompute(configs) "
                             "(or keep wf.updateinternals in step with configs.move) first")
        self._calls += 1
        out = dev.energy(self.threshold, rot=rot, unif=unif, seed=self.seed + self._calls)
        return {k: out[i] for i, k in enumerate(KEYS)}

    def avg(self, configs, wf):
        return {k: np.mean(v, axis=0) for k, v in self(configs, wf).items()}

    def has_nonlocal_moves(self):
        return self.mol._ecp != {}

    def keys(self):
        return set(KEYS)

    def shapes(self):
        return {k: () for k in KEYS}
