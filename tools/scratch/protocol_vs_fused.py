# Throughput of the drop-in protocol route (pyq.vmc unchanged: per-electron C calls + host accept/reject) against the fused
# route (pyqmc_amd.vmc -> pqa_vmc_sweeps) on the metric system.  python tools/scratch/protocol_vs_fused.py [walkers]
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import helpers
import numpy as np
import pyqmc_amd as pa
W = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
mol = pa.systems.water_cluster(); mf = pa.systems.random_mf(mol)
wf = pa.generate_wf(mol, mf)
acc = {"energy": pa.EnergyAccumulator(mol)}
for fused in (True, False):
    cfg = pa.initial_guess(mol, W, rng=np.random.default_rng(1))
    run = pa.vmc_worker if fused else helpers.protocol_vmc_worker
    run(wf, cfg, 0.3, 1, acc)
    n = 4 if fused else 1
    t0 = time.perf_counter(); run(wf, cfg, 0.3, n, acc); dt = time.perf_counter() - t0
    print("fused" if fused else "protocol", W, "walkers:", round(W * n / dt), "walker-steps/s", round(1e3 * dt / n, 1), "ms/step")
