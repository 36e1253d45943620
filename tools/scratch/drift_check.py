"""Round-off accumulated by Sherman-Morrison updates: log|Psi| carried through many sweeps vs a fresh recompute."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import pyqmc_amd as pa
from pyqmc_amd.configs import OpenConfigs
mol = pa.systems.water_cluster(); wf = pa.generate_wf(mol, pa.systems.random_mf(mol)); dev = wf.fused_device()
wf.recompute(pa.initial_guess(mol, 4096, rng=np.random.default_rng(1)))
for nsweep in (10, 100, 400):
    dev.vmc_sweeps(0.3, nsweep, seed=nsweep, energy=False)
    s0, l0 = wf.value()
    s1, l1 = wf.recompute(OpenConfigs(dev.configs()))
    print("sweeps", nsweep, "max |dlog|", float(np.max(np.abs(l0 - l1))), "sign flips", int((s0 != s1).sum()))
