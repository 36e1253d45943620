"""Sweep-only time of the three single-determinant sweeps of an open-boundary handle over shard sizes: the launch-per-move sweep
(PQA_RES=0), k_sweep_res (PQA_RES=1 PQA_R8=0: 16 walkers per 512-thread block) and k_sweep_r8 (PQA_RES=1 PQA_R8=1: 8 walkers per
256-thread block, two blocks per CU).  usage: r8_scan.py [M|C2] [W ...]"""
import os, sys, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
def run(system, W, env):
    os.environ.update(env)
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
system = sys.argv[1] if len(sys.argv) > 1 else "M"
Ws = [int(a) for a in sys.argv[2:]] or [4096, 16384, 32768, 65536]
for W in Ws:
    a = run(system, W, {"PQA_RES": "0", "PQA_R8": "0"})
    b = run(system, W, {"PQA_RES": "1", "PQA_R8": "0"})
    c = run(system, W, {"PQA_RES": "1", "PQA_R8": "1"})
    print(json.dumps(dict(system=system, W=W, launches_ms=round(a, 3), res16_ms=round(b, 3), r8_ms=round(c, 3))), flush=True)
