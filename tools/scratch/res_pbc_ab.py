"""C5's cell (2x2x2 diamond, 64 e-): VMC sweep-only and DMC step time with the resident sweep (PQA_RES=1) and without (PQA_RES=0)."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import pyqmc_amd as pa
from pyqmc_amd import pbc
sup = pbc.get_supercell(pa.systems.diamond_primitive(), 2.0 * np.eye(3))
mf = pbc.random_kmf(sup)
for W in [int(a) for a in sys.argv[1:]] or [4096]:
    wf = pa.generate_wf(sup, mf); dev = wf.fused_device()
    wf.recompute(pa.initial_guess(sup, W, rng=np.random.default_rng(11)))
    dev.vmc_sweeps(0.3, 2, seed=1, energy=False); dev.sync()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter(); dev.vmc_sweeps(0.3, 4, seed=2 + rep, energy=False); dev.sync()
        best = min(best, (time.perf_counter() - t0) / 4)
    w = np.ones(W)
    dev.dmc_steps(0.02, 2, w, 10.0, -40.0, -40.0, seed=3)
    t0 = time.perf_counter(); dev.dmc_steps(0.02, 6, w, 10.0, -40.0, -40.0, seed=4); dev.sync()
    print("PQA_RES", os.environ.get("PQA_RES"), W, "walkers: sweep ms", round(1e3 * best, 3), " dmc step ms", round(1e3 * (time.perf_counter() - t0) / 6, 3), flush=True)
