"""Sweep-only time of the walker-tile kernel for a given library build: python tile_time.py lib.so"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ["PQA_LW"] = os.environ.get("PQA_LW", "2")
from pyqmc_amd import _ffi
_ffi.LIB_PATH = os.path.abspath(sys.argv[1])
import numpy as np
import pyqmc_amd as pa
W = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
mol = pa.systems.water_cluster(); mf = pa.systems.random_mf(mol)
wf = pa.generate_wf(mol, mf); dev = wf.fused_device()
wf.recompute(pa.initial_guess(mol, W, rng=np.random.default_rng(1)))
dev.vmc_sweeps(0.3, 1, seed=1, energy=False); dev.sync()
t0 = time.perf_counter(); acc, _, _ = dev.vmc_sweeps(0.3, 3, seed=2, energy=False); dev.sync()
print(os.path.basename(sys.argv[1]), "sweep ms", round(1e3 * (time.perf_counter() - t0) / 3, 2), "acc", acc.mean())
