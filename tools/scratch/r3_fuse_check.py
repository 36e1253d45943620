"""Bitwise comparison of the one-launch-per-move sweep (PQA_LW_FUSE=1) with the six-launch sequence (PQA_LW_FUSE=0):
python r3_fuse_check.py [walkers]   (both handles in one process; the switch is read at pqa_create)"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import pyqmc_amd as pa

def run(fuse, mol, mf, W, steps=2, seed=5):
    os.environ["PQA_LW_FUSE"] = str(fuse)
    wf = pa.generate_wf(mol, mf); dev = wf.fused_device()
    cfg = pa.initial_guess(mol, W, rng=np.random.default_rng(1))
    wf.recompute(cfg)
    acc, en, _ = dev.vmc_sweeps(0.3, steps, seed=seed, energy=True)
    x = dev.configs()
    s, l = dev.value()
    return acc, en, x, l

for name, W in (("water_cluster", int(sys.argv[1]) if len(sys.argv) > 1 else 4096), ("water", 1000)):
    mol = getattr(pa.systems, name)(); mf = pa.systems.random_mf(mol)
    a0 = run(0, mol, mf, W); a1 = run(1, mol, mf, W)
    ok = all(np.array_equal(np.asarray(p), np.asarray(q)) for p, q in zip(a0, a1))
    print(name, W, "bitwise identical" if ok else "DIFFERENT", "acc", np.asarray(a1[0]).mean(),
          "max|dx|", float(np.max(np.abs(np.asarray(a0[2]) - np.asarray(a1[2])))), "max|dlog|", float(np.max(np.abs(a0[3] - a1[3]))))
