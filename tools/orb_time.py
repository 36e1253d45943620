"""Time only the orbital kernel of the metric system with a chosen build of the library (ablation studies).
   python tools/orb_time.py [libname.so] [npts]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyqmc_amd import _ffi
if len(sys.argv) > 1 and sys.argv[1] != "-":
    _ffi.LIB_PATH = os.path.join(os.path.dirname(_ffi.LIB_PATH), sys.argv[1])
import pyqmc_amd as pa
npts = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
mol = pa.systems.water_cluster(); mf = pa.systems.random_mf(mol)
wf = pa.generate_wf(mol, mf); dev = wf.fused_device()
cfg = pa.initial_guess(mol, npts, rng=np.random.default_rng(1))
wf.recompute(cfg)
dev.vmc_sweeps(0.3, 1, seed=1, energy=False); dev.sync()
dev.profile_enable(True)
dev.vmc_sweeps(0.3, 2, seed=2, energy=False); dev.sync()
n, ms, pc = dev.profile_query()
print(sys.argv[1] if len(sys.argv) > 1 else "default", "k_orb<5> avg us", round(1e3 * ms / n, 2), "launches", n)
