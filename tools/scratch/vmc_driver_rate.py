"""End-to-end rate of the production driver pa.vmc() (block loop, energies per block) on the headline system.
    python tools/scratch/vmc_driver_rate.py [walkers] [nblocks] [nsteps]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pyqmc_amd as pa

W = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 5
ns = int(sys.argv[3]) if len(sys.argv) > 3 else 10
mol = pa.systems.water_cluster()  # (H2O)8: 64 e-
wf = pa.generate_wf(mol, pa.systems.random_mf(mol))
cfg = pa.initial_guess(mol, W, rng=np.random.default_rng(1))
acc = {"energy": pa.EnergyAccumulator(mol)}
pa.vmc(wf, cfg, nblocks=1, nsteps_per_block=2, tstep=0.3, accumulators=acc, seed=1)
for every in (1, 10):
    t0 = time.perf_counter()
    df, cfg = pa.vmc(wf, cfg, nblocks=nb, nsteps_per_block=ns, tstep=0.3, accumulators=acc, seed=5, recompute_every=every)
    dt = time.perf_counter() - t0
    print(f"walkers {W}: {nb} blocks x {ns} steps, recompute_every={every}: {W * nb * ns / dt:.0f} walker-steps/s ({1e3 * dt / (nb * ns):.2f} ms/step), E = {np.mean(df['energytotal']):.4f}")
