"""Config C2 shape: H2O single-determinant Slater-Jastrow, VMC sweep + energy, walkers from argv."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import pyqmc_amd as pa
W = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
mol = pa.systems.water(); mf = pa.systems.random_mf(mol)
wf = pa.generate_wf(mol, mf); dev = wf.fused_device()
wf.recompute(pa.initial_guess(mol, W, rng=np.random.default_rng(1)))
dev.vmc_sweeps(0.3, 3, seed=1, energy=True); dev.sync()
K = 20
t0 = time.perf_counter(); acc, en, _ = dev.vmc_sweeps(0.3, K, seed=2, energy=True); dev.sync()
dt = time.perf_counter() - t0
print("LW", os.environ.get("PQA_LW", "1"), "W", W, "ms/step", round(1e3 * dt / K, 3), "walker-steps/s", round(W * K / dt), "acc", round(acc.mean(), 4), "E", round(en[:, 5].mean(), 4))
