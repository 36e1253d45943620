"""Trajectory of a fused sweep with a given library build -> npz (for bitwise comparison of two builds):
python r3_traj.py lib.so out.npz [walkers]; python r3_traj.py --cmp a.npz b.npz"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
if sys.argv[1] == "--cmp":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    for k in a.files:
        same = np.array_equal(a[k], b[k])
        print(k, "bitwise identical" if same else f"DIFFERENT max|d| {np.max(np.abs(a[k] - b[k])):.3e}")
    sys.exit(0)
from pyqmc_amd import _ffi
_ffi.LIB_PATH = os.path.abspath(sys.argv[1])
import pyqmc_amd as pa
out = {}
W = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
for name, Wn in (("water_cluster", W), ("water", 1000)):
    mol = getattr(pa.systems, name)(); mf = pa.systems.random_mf(mol)
    wf = pa.generate_wf(mol, mf); dev = wf.fused_device()
    wf.recompute(pa.initial_guess(mol, Wn, rng=np.random.default_rng(1)))
    acc, en, _ = dev.vmc_sweeps(0.3, 3, seed=5, energy=True)
    out[name + "_acc"], out[name + "_en"], out[name + "_x"], out[name + "_log"] = acc, en, dev.configs(), dev.value()[1]
np.savez(sys.argv[2], **out)
print("wrote", sys.argv[2], {k: float(np.mean(v)) for k, v in out.items() if k.endswith("_acc")})
