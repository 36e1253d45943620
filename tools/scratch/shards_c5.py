"""C5 (diamond 2x2x2 DMC, T-moves) with the walkers of ONE GPU split over S device handles, each driven by its own host thread
and HIP stream: at 4096 walkers every kernel of the step is a latency chain, so independent chains should overlap.
    python tools/scratch/shards_c5.py [walkers] [steps]"""
import os, sys, time, threading
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pyqmc_amd as pa
from pyqmc_amd import pbc

W = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
K = int(sys.argv[2]) if len(sys.argv) > 2 else 10
sup = pbc.get_supercell(pa.systems.diamond_primitive(), 2.0 * np.eye(3))
for S in (1, 2, 4):
    shards = []
    for i in range(S):
        wf = pa.generate_wf(sup, pbc.random_kmf(sup))
        cfg = pa.initial_guess(sup, W // S, rng=np.random.default_rng(1 + i))
        wf.recompute(cfg)
        acc = {"energy": pa.EnergyAccumulator(sup)}
        w = np.ones(W // S)
        pa.dmc_propagate(wf, cfg, w, 0.02, 3.0, -40.0, -40.0, nsteps=2, accumulators=acc)
        shards.append([wf, cfg, w, acc])
    def run(sh):
        sh[0].fused_device().sync()
        blk, sh[1], sh[2] = pa.dmc_propagate(sh[0], sh[1], sh[2], 0.02, 3.0, -40.0, -40.0, nsteps=K, accumulators=sh[3])
    t0 = time.perf_counter()
    th = [threading.Thread(target=run, args=(sh,)) for sh in shards]
    [t.start() for t in th]
    [t.join() for t in th]
    dt = time.perf_counter() - t0
    print(f"walkers {W} in {S} shard(s): {1e3 * dt / K:.2f} ms/step, {W * K / dt:.0f} walker-steps/s", flush=True)
    del shards
