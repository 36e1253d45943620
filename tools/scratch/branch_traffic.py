"""How many walkers does the sorted stochastic comb send across ranks?  

One device handle holds the whole C5 ensemble (diamond 2x2x2, DMC tstep 0.02, T-moves); after each block of 5 fused steps the
weights are those a G-rank run would all-gather, so `dist.exchange_plan` on contiguous shards of W/G walkers tells how many
walkers would travel for G = 2, 4, 8 — with the comb's survivor list sorted (what branch_distributed does) and in the
reference's cyclic order.

    python tools/scratch/branch_traffic.py [--walkers 32768] [--blocks 4]
"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pyqmc_amd as pa  # noqa: E402
from pyqmc_amd import dist as pdist, dmc, pbc  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--walkers", type=int, default=32768)
ap.add_argument("--blocks", type=int, default=4)
a = ap.parse_args()

W = a.walkers
sup = pbc.get_supercell(pa.systems.diamond_primitive(), 2.0 * np.eye(3))
wf = pa.generate_wf(sup, pbc.random_kmf(sup))
cfg = pa.initial_guess(sup, W, rng=np.random.default_rng(1))
wf.recompute(cfg)
dev = wf.fused_device()
dev.vmc_sweeps(0.3, 3, seed=1, energy=False)
cfg.configs[...] = dev.configs()
acc = {"energy": pa.EnergyAccumulator(sup)}
weights = np.ones(W)
blk, cfg, weights = pa.dmc_propagate(wf, cfg, weights, 0.02, 3.0, -40.0, -40.0, nsteps=2, accumulators=acc)
eref = float(np.real(blk["energytotal"]))
weights = np.ones(W)
rng = np.random.default_rng(7)
bytes_per_walker = cfg.configs[0].size * 8 + (cfg.wrap[0].size * 8 if hasattr(cfg, "wrap") else 0)
for b in range(a.blocks):
    blk, cfg, weights = pa.dmc_propagate(wf, cfg, weights, 0.02, 3.0, eref, eref, nsteps=5, accumulators=acc)
    u = float(rng.random())
    newinds = dmc.comb_indices(weights, u)[0]
    row = {"block": b, "walkers": W, "weight_std_over_mean": float(np.std(weights) / np.mean(weights)),
           "distinct_survivors": int(len(np.unique(newinds))), "bytes_per_walker": int(bytes_per_walker)}
    for G in (2, 4, 8):
        counts = [W // G] * G
        for name, inds in (("sorted", np.sort(newinds)), ("cyclic", newinds)):
            moved = sum(sum(pdist.exchange_plan(inds, counts, r)[1].values()) for r in range(G))
            row[f"moved_{name}_G{G}"] = int(moved)
    print(json.dumps(row), flush=True)
    # branch on the device like rundmc does, so the next block starts from the combed ensemble
    cfg.resample(newinds)
    wf.recompute(cfg)
    weights = np.full(W, np.mean(weights))
