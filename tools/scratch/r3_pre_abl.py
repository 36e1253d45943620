"""k_pbc_prepass timing with an ablated library: python r3_pre_abl.py lib.so walkers (timing only: the lists are wrong)"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from pyqmc_amd import _ffi
_ffi.LIB_PATH = os.path.abspath(sys.argv[1])
import numpy as np
import pyqmc_amd as pa
from pyqmc_amd import pbc, systems
W = int(sys.argv[2])
sup = pbc.get_supercell(systems.diamond_primitive(), 2.0 * np.eye(3)); mf = pbc.random_kmf(sup)
wf = pa.generate_wf(sup, mf); dev = wf.fused_device()
wf.recompute(pa.initial_guess(sup, W, rng=np.random.default_rng(1)))
dev.vmc_sweeps(0.3, 1, seed=1, energy=False); dev.sync()
t0 = time.perf_counter(); dev.vmc_sweeps(0.3, 3, seed=2, energy=False); dev.sync()
print(os.path.basename(sys.argv[1]), W, "sweep ms", round(1e3 * (time.perf_counter() - t0) / 3, 3))
