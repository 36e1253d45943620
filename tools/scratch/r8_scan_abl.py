import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
os.environ["PQA_RES"]="1"
import pyqmc_amd as pa
mol = pa.systems.water_cluster(); mf = pa.systems.random_mf(mol)
for W in (16384, 65536):
    wf = pa.generate_wf(mol, mf); dev = wf.fused_device()
    wf.recompute(pa.initial_guess(mol, W, rng=np.random.default_rng(11)))
    dev.vmc_sweeps(0.3, 2, seed=1, energy=False); dev.sync()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter(); dev.vmc_sweeps(0.3, 4, seed=2 + rep, energy=False); dev.sync()
        best = min(best, (time.perf_counter() - t0) / 4)
    print(os.environ.get("PQA_R8_ABL"), W, "sweep ms", round(1e3 * best, 3), flush=True)
