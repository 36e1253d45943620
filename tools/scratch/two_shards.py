"""A/B: one handle with W walkers vs S handles with W/S walkers each driven by S host threads (own HIP streams)."""
import sys, os, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import bench
import pyqmc_amd as pa

W = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
K = 8
for S in (1, 2, 4):
    shards = []
    for i in range(S):
        mol, mf, wf = bench.build_wf(0)
        dev = wf.fused_device()
        cfg = pa.initial_guess(mol, W // S, rng=np.random.default_rng(1234 + i))
        wf.recompute(cfg)
        dev.vmc_sweeps(0.3, 2, seed=5 + i, energy=True)
        shards.append((wf, dev))
    for _, d in shards:
        d.sync()
    def run(d, i):
        d.vmc_sweeps(0.3, K, seed=99 + i, energy=True)
        d.sync()
    t0 = time.perf_counter()
    th = [threading.Thread(target=run, args=(d, i)) for i, (_, d) in enumerate(shards)]
    [t.start() for t in th]
    [t.join() for t in th]
    dt = time.perf_counter() - t0
    print(f"shards {S}: {1e3 * dt / K:.2f} ms/step, {W * K / dt:.0f} walker-steps/s", flush=True)
    del shards
