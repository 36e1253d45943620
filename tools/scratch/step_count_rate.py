"""ms/step of the fused sweep + energy as a function of the steps per call and of what ran before (is the per-step cost
independent of the call length / of a sweep-only call in between?)
    python tools/scratch/step_count_rate.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pyqmc_amd as pa

W = 65536
mol = pa.systems.water_cluster()
wf = pa.generate_wf(mol, pa.systems.random_mf(mol))
wf.recompute(pa.initial_guess(mol, W, rng=np.random.default_rng(1)))
dev = wf.fused_device()
dev.vmc_sweeps(0.3, 3, seed=7, energy=True); dev.sync()

def run(steps, energy, tag):
    t0 = time.perf_counter()
    acc, en, _ = dev.vmc_sweeps(0.3, steps, seed=6 + steps, energy=energy)
    dev.sync()
    dt = time.perf_counter() - t0
    print(f"{tag} steps {steps:3d} energy {energy}: {1e3 * dt / steps:.2f} ms/step  acc {float(np.mean(acc)):.4f}", flush=True)

for rep in range(3): run(8, True, "A")
run(5, False, "B")
for rep in range(3): run(8, True, "C")
run(5, False, "B")
for rep in range(3): run(4, True, "D")
for rep in range(3): run(8, True, "E")
