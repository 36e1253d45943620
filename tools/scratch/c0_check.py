"""Per-walker energies of the headline system after NS sweeps: saves [6][W] to the file given (compare two builds: PQA_LIB)."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import pyqmc_amd as pa
out, W, NS = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
if os.path.exists(out):
    a = np.load(out); b = np.load(sys.argv[4])
    d = np.abs(a - b)
    print("rows ke ee ei ecp grad2 total: max abs diff", d.max(axis=1), "\nke means of the last steps", a[0][-3:], b[0][-3:], "\nrel diff of the steps' ke means", (d[0] / np.abs(a[0])))
    sys.exit(0)
mol = pa.systems.water_cluster(); mf = pa.systems.random_mf(mol)
wf = pa.generate_wf(mol, mf); dev = wf.fused_device()
wf.recompute(pa.initial_guess(mol, W, rng=np.random.default_rng(11)))
# large shards only take the quad kernel (>= 16384 walkers)
acc, en, rec = dev.vmc_sweeps(0.3, NS, seed=21, energy=True, record=False)
np.save(out, np.asarray(en).T)  # [6][NS]: the steps' means over the walkers (the fused pass: k_kinetic_lw)
