"""Headline step time with a given library build: python lib_bench.py lib.so [walkers]"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from pyqmc_amd import _ffi
_ffi.LIB_PATH = os.path.abspath(sys.argv[1])
import numpy as np
import pyqmc_amd as pa
W = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
mol = pa.systems.water_cluster(); mf = pa.systems.random_mf(mol)
wf = pa.generate_wf(mol, mf); dev = wf.fused_device()
wf.recompute(pa.initial_guess(mol, W, rng=np.random.default_rng(1)))
dev.vmc_sweeps(0.3, 2, seed=1, energy=True); dev.sync()
best = 1e9
for rep in range(3):
    t0 = time.perf_counter(); dev.vmc_sweeps(0.3, 4, seed=2 + rep, energy=True); dev.sync()
    best = min(best, (time.perf_counter() - t0) / 4)
print(os.path.basename(sys.argv[1]), "ms/step", round(1e3 * best, 2))
