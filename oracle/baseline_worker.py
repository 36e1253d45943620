"""TEST / MEASUREMENT INFRASTRUCTURE — one worker of the CPU baseline (bench.py's ``cpu_baseline`` leg only).

SURVEY.md section 8(d) / BASELINE.md section 3: the reference scales by independent single-thread worker processes, each with
its share of the walkers and no communication inside a block (``vmc_parallel``, pyqmc/method/mc.py:156-173).  The baseline
therefore runs P = physical cores of one socket concurrent processes, each executing the NumPy oracle (reference structure:
two ``gradient_value`` per move, per-(electron, atom) ECP loop, energy after every sweep) on its own walkers.  This module is
what each spawned process runs: it imports numpy and the oracle only (no torch, no HIP).
"""

import os
import time


def run(args):
    """args = (index, walkers, nsteps, tstep, cpu or None, start_at[, ao_backend]).  Builds the wave function and walkers, waits
    for the common start time, runs the oracle's vmc_worker; returns (index, t_begin, t_end, walker_steps, ao_seconds).
    ao_backend "c" (default): AOs from oracle/libao_eval.so — a compiled single-thread routine, as the reference's default AO
    back end is (orbitals.py:46-51); "numpy": the oracle's vectorised NumPy routine."""
    idx, walkers, nsteps, tstep, cpu, start_at = args[:6]
    ao_backend = args[6] if len(args) > 6 else "c"
    for v in ("OMP_NUM_THREADS", "MKL_NUM_THREADS", "OPENBLAS_NUM_THREADS"):
        os.environ[v] = "1"
    if cpu is not None:
        try:
            os.sched_setaffinity(0, {cpu})
        except OSError:
            pass
    import numpy as np

    import pyqmc_amd as pa
    from oracle import gto as ogto
    from oracle import vmc as ovmc
    from tests import helpers

    ogto.set_ao_backend(ao_backend)

    mol = pa.systems.water_cluster()
    mf = pa.systems.random_mf(mol)
    owf = helpers.oracle_wf(mol, mf)
    rng = np.random.default_rng(5 + idx)
    cfg = pa.initial_guess(mol, walkers, rng=rng)
    N, necp = int(sum(mol.nelec)), mol.natm
    gauss, unif = rng.standard_normal((nsteps, N, walkers, 3)), rng.random((nsteps, N, walkers))
    rot = np.broadcast_to(np.eye(3), (nsteps, N, necp, 3, 3)).copy()
    eunif = rng.random((nsteps, N, necp, walkers))
    owf.recompute(cfg)  # set-up stays outside the clock, as the reference's vmc_worker starts from a recompute too
    while time.time() < start_at:  # all workers start together: the cores contend for memory bandwidth as in a production run
        time.sleep(0.01)
    ogto.AO_SECONDS = 0.0
    t0 = time.time()
    ovmc.vmc_worker(mol, owf, cfg, tstep, gauss, unif, rot, eunif)
    return idx, t0, time.time(), walkers * nsteps, ogto.AO_SECONDS
