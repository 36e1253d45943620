"""Sweep-only time of the resident sweep (PQA_RES=1) and the launch-per-move sweep (PQA_RES=0) over shard sizes."""
import os, sys, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
def run(system, W, res):
    os.environ["PQA_RES"] = str(res)
    import pyqmc_amd as pa
    mol = pa.systems.water_cluster() if system == "M" else pa.systems.water()
    mf = pa.systems.random_mf(mol)
    wf = pa.generate_wf(mol, mf); dev = wf.fused_device()
    wf.recompute(pa.initial_guess(mol, W, rng=np.random.default_rng(11)))
    dev.vmc_sweeps(0.3, 2, seed=1, energy=False); dev.sync()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter(); dev.vmc_sweeps(0.3, 4, seed=2 + rep, energy=False); dev.sync()
        best = min(best, (time.perf_counter() - t0) / 4)
    return 1e3 * best
for system, Ws in (("M", [512, 1024, 2048, 4096, 8192, 12288, 16384, 24576, 32768, 49152, 65536]), ("C2", [1024, 4096, 16384])):
    for W in Ws:
        a, b = run(system, W, 0), run(system, W, 1)
        print(json.dumps(dict(system=system, W=W, lw_ms=round(a, 3), res_ms=round(b, 3), ratio=round(a / b, 3))), flush=True)
