#!/bin/bash
# A/B of library builds on the resident sweep: bash tools/scratch/res_ab.sh lib1.so lib2.so ... (each: sweep-only ms at 4096 and 65536 walkers)
for l in "$@"; do
  echo "== $l"
  PQA_LIB=$l PQA_RES=1 timeout 300 python - <<'PY'
import os, sys, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import pyqmc_amd as pa
mol = pa.systems.water_cluster(); mf = pa.systems.random_mf(mol)
for W in (4096, 65536):
    wf = pa.generate_wf(mol, mf); dev = wf.fused_device()
    wf.recompute(pa.initial_guess(mol, W, rng=np.random.default_rng(11)))
    dev.vmc_sweeps(0.3, 2, seed=1, energy=False); dev.sync()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter(); dev.vmc_sweeps(0.3, 4, seed=2 + rep, energy=False); dev.sync()
        best = min(best, (time.perf_counter() - t0) / 4)
    print(W, "sweep ms", round(1e3 * best, 3), flush=True)
PY
done
