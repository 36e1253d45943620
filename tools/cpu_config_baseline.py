"""CPU baselines of the BASELINE.json configurations other than the headline one.

    python tools/cpu_config_baseline.py [c2 c3 c4 c5 big] [--procs P]

Same model as bench.py's `cpu_baseline`: P concurrent single-thread processes (P = the CPUs the container may use, at most the
physical cores of one socket), each running the NumPy ORACLE — reference algorithm and structure — on its own walkers; all start
together; rate = total walker-steps / (last end - first start).  AOs come from the compiled routines of oracle/ao_eval.c — molecular
(`ao_eval`) and, since round 4, periodic lattice sums (`ao_eval_pbc`: pbcgto.py:99-506 compiled, as the reference's default
periodic back end is compiled code) — `ao_share_of_wall_time` says how much of the time they are, `non_ao_rate_per_core` gives
the rate with the AO time taken out (`--ao numpy` reproduces the round-3 lines with NumPy lattice sums).  One JSON line per configuration."""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SIZES = {"c2": (2048, 4), "c3": (256, 2), "c4": (256, 4), "c5": (64, 1), "big": (8, 1)}  # (walkers per process, steps): a few seconds of work each


def worker(args):
    name, idx, cpu, start_at, ao = args
    for v in ("OMP_NUM_THREADS", "MKL_NUM_THREADS", "OPENBLAS_NUM_THREADS"):
        os.environ[v] = "1"
    if cpu is not None:
        try:
            os.sched_setaffinity(0, {cpu})
        except OSError:
            pass
    import numpy as np

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import helpers
    import pyqmc_amd as pa
    from oracle import dmc as odmc
    from oracle import gto as ogto
    from oracle import vmc as ovmc
    from pyqmc_amd import pbc, systems
    from helpers import NumpyRNG as _NumpyRNG

    ogto.set_ao_backend(ao)
    W, nsteps = SIZES[name]
    rng = np.random.default_rng(5 + idx)
    np.random.seed(5 + idx)
    if name == "big":  # (H2O)18: 72 + 72 electrons, 414 AOs (tools/config_bench.py big)
        mol = systems.water_cluster(3, 3, 2)
        wf = helpers.oracle_wf(mol, systems.random_mf(mol))
    elif name in ("c2", "c4"):
        mol = systems.water()
        if name == "c2":
            wf = helpers.oracle_wf(mol, systems.random_mf(mol))
        else:
            mf = systems.random_mf(mol, nvirt=8)
            wf = helpers.oracle_wf3(mol, mf, systems.random_determinants(mol, mf, 50))
    else:
        S = np.array([[-1.0, 1, 1], [1, -1, 1], [1, 1, -1]]) if name == "c3" else 2.0 * np.eye(3)
        mol = pbc.get_supercell(systems.diamond_primitive(), S)
        mf = pbc.random_kmf(mol, complex_coeff=True, twist=(0.25, 0.1, -0.3)) if name == "c3" else pbc.random_kmf(mol)
        from oracle import jastrow_basis, wf as owf

        Ls = pbc.lattice_points_within(mol.original_cell.lattice_vectors(), 30.0)
        sl = owf.Slater.periodic(mol, mf.kpts, mf.mo_coeff, Ls)
        rcut = float(np.amin(np.pi / np.linalg.norm(mol.reciprocal_vectors(), axis=1)))
        ab, bb, rcut = jastrow_basis.default_basis(ion_cusp=False, rcut=rcut)
        ja = owf.JastrowSpin(mol, ab, bb, rcut)
        ja.parameters["acoeff"], ja.parameters["bcoeff"] = helpers.pbc_jastrow_coeffs(mol)
        wf = owf.MultiplyWF(sl, ja)
    cfg = pa.initial_guess(mol, W, rng=rng)
    N, necp = int(sum(mol.nelec)), mol.natm
    gauss, unif = rng.standard_normal((nsteps, N, W, 3)), rng.random((nsteps, N, W))
    rot = np.broadcast_to(np.eye(3), (nsteps, N, necp, 3, 3)).copy()
    eunif = rng.random((nsteps, N, necp, W))
    wf.recompute(cfg)
    while time.time() < start_at:
        time.sleep(0.01)
    ogto.AO_SECONDS = 0.0
    t0 = time.time()
    if name == "c5":  # DMC step with T-moves (dmc.py:123-221), tstep 0.02
        odmc.dmc_propagate(mol, wf, cfg, np.ones(W), 0.02, 10.0, -40.0, -40.0, nsteps, _NumpyRNG())
    else:
        ovmc.vmc_worker(mol, wf, cfg, 0.3, gauss, unif, rot, eunif)
    return name, t0, time.time(), W * nsteps, ogto.AO_SECONDS


def main():
    sys.path.insert(0, ROOT)
    import bench

    ap = argparse.ArgumentParser()
    ap.add_argument("configs", nargs="*", default=["c2", "c3", "c4", "c5"])
    ap.add_argument("--procs", type=int, default=0)
    ap.add_argument("--ao", default="c", choices=["c", "numpy"])
    a = ap.parse_args()
    cpus, model = bench.socket_cores()
    socket = len(cpus)
    quota = bench.cgroup_cpu_quota()
    if quota and quota < len(cpus):
        cpus = cpus[: int(quota)]
    if a.procs > 0:
        cpus = cpus[: a.procs]
    P = len(cpus)
    ctx = mp.get_context("spawn")
    for name in a.configs:
        start_at = time.time() + 25.0 + 0.05 * P
        with ctx.Pool(P) as pool:
            res = pool.map(worker, [(name, i, cpus[i], start_at, a.ao) for i in range(P)], chunksize=1)
        t_begin, t_end = min(r[1] for r in res), max(r[2] for r in res)
        per_core = sum(r[3] / (r[2] - r[1]) for r in res) / P
        print(json.dumps({"config": name, "value": sum(r[3] for r in res) / (t_end - t_begin), "unit": "walker-steps/s", "cores": P, "per_core": per_core,
                          "socket_extrapolated": per_core * socket, "socket_physical_cores": socket, "cpu_model": model, "kind": "port",
                          "ao_backend": ("c (oracle/ao_eval.c: " + ("ao_eval" if name in ("c2", "c4") else "ao_eval_pbc") + ", gcc -O3)") if a.ao == "c" else "numpy",
                          "ao_share_of_wall_time": sum(r[4] for r in res) / sum(r[2] - r[1] for r in res),
                          "non_ao_rate_per_core": sum(r[3] / max(r[2] - r[1] - r[4], 1e-9) for r in res) / P,
                          "sample": f"{P} processes x {SIZES[name][0]} walkers x {SIZES[name][1]} steps in {t_end - t_begin:.1f} s; "
                                    + ("DMC step incl. T-moves, tstep 0.02" if name == "c5" else "VMC sweep + energy, tstep 0.3")}), flush=True)


if __name__ == "__main__":
    main()
