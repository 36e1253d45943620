"""Throughput of the BASELINE.json configurations other than the headline one (synthetic tables, see pyqmc_amd.systems).

    python tools/config_bench.py c2|c3|c4|c5|big|big_pbc [--walkers W] [--steps K]

c2: H2O single-determinant Slater-Jastrow VMC, 4096 walkers (lane-per-walker sweep; launch-latency bound at this size)

c3: diamond conventional cell (8 atoms, 32 e-) with a k-point twist, Slater-Jastrow VMC (complex wave-per-walker sweep)
c4: H2O, 50 determinants x 2-body x 3-body Jastrow, VMC (wave-per-walker sweep)
c5: diamond 2x2x2 supercell (64 e-, 8 k-points), DMC tstep = 0.02 (pqa_dmc_steps; --host: dmc_propagate over the protocol)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))  # --host: the protocol-route harness
import pyqmc_amd as pa  # noqa: E402
import helpers  # noqa: E402
from pyqmc_amd import pbc  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("config")
ap.add_argument("--walkers", type=int, default=0)
ap.add_argument("--steps", type=int, default=2)
ap.add_argument("--rundmc", type=int, default=0, help="c5: time rundmc blocks (5 steps + branching each) with this recompute_every instead of bare steps")
ap.add_argument("--repeat", type=int, default=2, help="timed passes; the fastest is reported")
ap.add_argument("--host", action="store_true", help="c5: drive the DMC step from the host over the protocol entry points")
a = ap.parse_args()
prim = pa.systems.diamond_primitive()
if a.config == "c2":
    W = a.walkers or 4096
    sup = mol = pa.systems.water()
    wf = pa.generate_wf(mol, pa.systems.random_mf(mol))
    cfg = pa.initial_guess(mol, W, rng=np.random.default_rng(1))
elif a.config == "c3":
    W = a.walkers or 8192
    sup = pbc.get_supercell(prim, np.array([[-1.0, 1, 1], [1, -1, 1], [1, 1, -1]]))
    mf = pbc.random_kmf(sup, complex_coeff=True, twist=(0.25, 0.1, -0.3))
    wf = pa.generate_wf(sup, mf)
    cfg = pa.initial_guess(sup, W, rng=np.random.default_rng(1))
elif a.config == "c4":
    W = a.walkers or 2048
    mol = pa.systems.water()
    mf = pa.systems.random_mf(mol, nvirt=8)
    wf = pa.generate_wf(mol, mf, determinants=pa.systems.random_determinants(mol, mf, 50), jastrow3=True)
    wf.parameters["wf3ccoeff"] = 0.05 * np.random.default_rng(2).standard_normal(wf.parameters["wf3ccoeff"].shape)
    cfg = pa.initial_guess(mol, W, rng=np.random.default_rng(1))
    sup = mol
elif a.config == "c5":
    W = a.walkers or 4096
    sup = pbc.get_supercell(prim, 2.0 * np.eye(3))
    wf = pa.generate_wf(sup, pbc.random_kmf(sup))
    cfg = pa.initial_guess(sup, W, rng=np.random.default_rng(1))
elif a.config == "big":  # (H2O)18: 72 + 72 electrons, 414 AOs — beyond 64 per spin (general-n kernels)
    W = a.walkers or 1024
    sup = mol = pa.systems.water_cluster(3, 3, 2)
    wf = pa.generate_wf(mol, pa.systems.random_mf(mol))
    cfg = pa.initial_guess(mol, W, rng=np.random.default_rng(1))
elif a.config == "big_pbc":  # diamond, 2x2x2 conventional cells as one cell at Gamma: 64 atoms, 128 + 128 electrons, 832 AOs, real orbitals
    W = a.walkers or 256
    sup = pbc.get_supercell(pa.systems.diamond_cubic(2), np.eye(3))
    wf = pa.generate_wf(sup, pbc.random_kmf(sup))
    cfg = pa.initial_guess(sup, W, rng=np.random.default_rng(1))
elif a.config == "big_complex":  # diamond 3x3x3 primitive cells: 54 atoms, 108 + 108 electrons, complex Bloch orbitals (27 k-points)
    W = a.walkers or 128
    sup = pbc.get_supercell(prim, 3.0 * np.eye(3))
    wf = pa.generate_wf(sup, pbc.random_kmf(sup))
    cfg = pa.initial_guess(sup, W, rng=np.random.default_rng(1))
else:
    raise SystemExit("config must be c2, c3, c4, c5, big, big_pbc or big_complex")
dev = wf.fused_device()
wf.recompute(cfg)
if a.config == "c5" and a.rundmc:
    acc = {"energy": pa.EnergyAccumulator(sup)}
    np.random.seed(1)
    nb = max(a.steps, 2)
    pa.rundmc(wf, cfg, tstep=0.02, nblocks=1, nsteps_per_block=5, vmc_warmup=1, accumulators=acc, recompute_every=a.rundmc)
    t0 = time.perf_counter()
    df, cfg, weights = pa.rundmc(wf, cfg, tstep=0.02, nblocks=nb, nsteps_per_block=5, vmc_warmup=0, accumulators=acc, recompute_every=a.rundmc)
    dt = time.perf_counter() - t0
    a.steps = 5 * nb
    kind = f"rundmc blocks: 5 steps + stochastic comb (recompute every {a.rundmc} blocks)"
    extra = {"acceptance": float(df["acceptance"].mean()), "tmove_acceptance": float(df["tmove_acceptance"].mean()), "weight": float(df["weight"].mean())}
elif a.config == "c5":
    acc = {"energy": pa.EnergyAccumulator(sup)}
    weights = np.ones(W)
    (helpers.protocol_dmc_propagate if a.host else pa.dmc_propagate)(wf, cfg, weights, 0.02, 3.0, -40.0, -40.0, nsteps=2, accumulators=acc)
    dt = float("inf")
    for _ in range(1 if a.host else a.repeat):  # best of `repeat` timed passes: about one process in eight sees a 1.5-2x slow pass
        t0 = time.perf_counter()
        blk, cfg, weights = (helpers.protocol_dmc_propagate if a.host else pa.dmc_propagate)(wf, cfg, weights, 0.02, 3.0, -40.0, -40.0, nsteps=a.steps, accumulators=acc)
        dt = min(dt, time.perf_counter() - t0)
    kind = "DMC (host-driven protocol path)" if a.host else "DMC (pqa_dmc_steps)"
    extra = {k: float(np.real(blk[k])) for k in ("acceptance", "tmove_acceptance", "weight")}
else:
    dev.vmc_sweeps(0.3, 2, seed=1, energy=True)  # two warm-up steps: first launches load code objects, the tile-width tuner samples
    dev.sync()
    dt = float("inf")
    for rep in range(a.repeat):  # best of `repeat` timed passes
        t0 = time.perf_counter()
        dev.vmc_sweeps(0.3, a.steps, seed=2 + rep, energy=True)
        dev.sync()
        dt = min(dt, time.perf_counter() - t0)
    kind = "VMC fused sweep + energy"
    extra = {}
if a.config == "big":
    # Roofline of the step for the open-boundary single-determinant handle beyond 64 electrons per spin (round-5 verdict item 6 iii): useful fp64 flops of
    # a walker-step by SURVEY 8(d)'s formula sheet — per move F_ao(5) = 30 P + 20 M, the contraction 2*5*M*n, the ratio sums 16 n, two Jastrow evaluations of
    # ~110 flops per pair, Sherman-Morrison 4 n^2 per accepted move (acceptance taken as 0.5); per electron of the energy 10 n + 1.3 Jastrow evaluations; the ECP
    # points are left out (a lower bound of the work) — against the 78.6 TFLOP/s fp64 pipe.
    from pyqmc_amd import tables

    t = tables.basis_tables(sup)
    N, n, M, P = int(sum(sup.nelec)), int(max(sup.nelec)), int(t["nao"]), int(len(t["prim_exp"]))
    fj = 110.0 * (N - 1 + sup.natm)
    f_step = N * (30 * P + 20 * M + 2 * 5 * M * n + 16 * n + 2 * fj + 0.5 * 4 * n * n) + N * (10 * n + 1.3 * fj)
    ach = f_step * W * a.steps / dt / 1e12
    extra["roofline"] = {"bound": "mfma", "useful_flop_per_walker_step": f_step, "achieved": ach, "peak": 78.6, "unit": "TFLOP/s", "frac": ach / 78.6,
                         "note": "wave-per-walker kernels (k_propose / k_accept per move, DESIGN 16.2): a correctness path, this is how far from the pipe it runs"}
print(json.dumps({"config": a.config, "kind": kind, "nelec": int(sum(sup.nelec)), "walkers": W, "steps": a.steps,
                  "ms_per_step": 1e3 * dt / a.steps, "walker_steps_per_s": W * a.steps / dt, **extra}))
