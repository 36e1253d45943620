# Test tool: time the fused sweep with the current build ("new") or with a second build placed at pyqmc_amd/lib/libpyqmc_amd_old.so ("old").
# A/B of two builds of the library on the same box (old build lacks the newest entry points: bind what exists)
import os, sys, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from pyqmc_amd import _ffi
which = sys.argv[1]
if which == "old":
    _ffi.LIB_PATH = _ffi.LIB_PATH.replace("libpyqmc_amd.so", "libpyqmc_amd_old.so")
    for k in ("pqa_set_ewald", "pqa_get_wrap"):
        _ffi._PROTOTYPES.pop(k)
import time, numpy as np
import pyqmc_amd as pa
mol = pa.systems.water_cluster(); mf = pa.systems.random_mf(mol)
wf = pa.generate_wf(mol, mf); dev = wf.fused_device()
cfg = pa.initial_guess(mol, 32768, rng=np.random.default_rng(1))
wf.recompute(cfg)
dev.vmc_sweeps(0.3, 2, seed=1, energy=True); dev.sync()
best = 1e9
for rep in range(3):
    t0 = time.perf_counter(); dev.vmc_sweeps(0.3, 4, seed=2 + rep, energy=True); dev.sync()
    best = min(best, (time.perf_counter() - t0) / 4)
print(which, "ms/step", round(1e3 * best, 3))
