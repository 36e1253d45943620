"""Calibration of bench.py's CPU baseline (BUILD CONTAINER ONLY: imports the real reference from /root/reference).

SURVEY.md section 8(d) / BASELINE.md section 3.2: the CPU number reported beside the GPU number is the NumPy oracle's, because the
reference cannot travel to the GPU box.  This script measures, on identical inputs, the wall time of the REAL reference's
`vmc_worker` (sweep + EnergyAccumulator per step) against the oracle's for the metric system (64-electron (H2O)8).  The
reference's default AO evaluator is PySCF's compiled libcgto and its numba evaluator is compiled too; neither exists here
(numba is stubbed to the identity, so `numba/gto.py` runs as interpreted Python), so the AO part cannot be timed
meaningfully.  Both sides are therefore given the SAME AO routine (the oracle's vectorised NumPy evaluator), which makes
the ratio a statement about everything else: determinant algebra, Jastrow, ECP loop, energy — the reference's own NumPy.

    python tools/calibrate_cpu_baseline.py [walkers] [steps]     -> one JSON line (paste into BASELINE.md section 3.2)
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
for v in ("OMP_NUM_THREADS", "MKL_NUM_THREADS", "OPENBLAS_NUM_THREADS"):
    os.environ[v] = "1"
import make_golden as mg  # noqa: E402  (stubs numba / pyscf / h5py and imports the reference)
import numpy as np  # noqa: E402

import helpers  # noqa: E402
import pyqmc.wf.orbitals as reforb  # noqa: E402
from oracle import gto as ogto, vmc as ovmc  # noqa: E402
from pyqmc_amd import systems  # noqa: E402


def main():
    W = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    mol = systems.water_cluster()
    mf = systems.random_mf(mol)
    table = ogto.AOTable(mol)
    comp = {"GTOval_sph": 1, "GTOval_sph_deriv1": 4, "GTOval_sph_deriv2": 5}

    def aos(self, eval_str, configs, mask=None):  # the oracle's AO routine behind the reference's evaluator interface (orbitals.py:85-93)
        x = configs.configs if mask is None else configs.configs[mask]
        nc = comp[eval_str]
        ao = ogto.eval_ao(table, x.reshape(-1, 3), nc)  # (nc, npts, nao), points flattened as the reference does
        return ao[0][None] if nc == 1 else ao[None]

    reforb.MoleculeOrbitalEvaluator.aos = aos
    rwf = mg.make_wf(mol, mf)
    owf = helpers.oracle_wf(mol, mf)
    start = mg.walkers(mol, W, 3)
    N, natm = int(sum(mol.nelec)), mol.natm
    # reference: its own vmc_worker with an energy accumulator, draws routed through recorded tapes
    with mg.Tapes(5) as t:
        t0 = time.perf_counter()
        blk, _ = mg.vmc_worker(rwf, mg.OpenConfigs(start.configs.copy()), 0.3, nsteps, {"energy": mg.pyq.EnergyAccumulator(mol)})
        t_ref = time.perf_counter() - t0
    gauss = np.asarray(t.log["normal"]).reshape(nsteps, N, W, 3)
    unif = np.asarray(t.log["rand"]).reshape(nsteps, N, W)
    rot = np.asarray(t.log["rot"]).reshape(nsteps, N, natm, 3, 3)
    eunif = np.asarray(t.log["random"]).reshape(nsteps, N, natm, W)
    t0 = time.perf_counter()
    oblk, _ = ovmc.vmc_worker(mol, owf, mg.OpenConfigs(start.configs.copy()), 0.3, gauss, unif, rot, eunif)
    t_orc = time.perf_counter() - t0
    de = abs(oblk["energytotal"] - blk["energytotal"]) / abs(blk["energytotal"])
    print(json.dumps({"system": "(H2O)8 64 e-", "walkers": W, "steps": nsteps, "reference_s": t_ref, "oracle_s": t_orc,
                      "ratio_oracle_over_reference": t_orc / t_ref, "rel_energy_difference": de,
                      "reference_move_s": float(blk["move time"]), "reference_accumulator_s": float(blk["accumulator time"]),
                      "oracle_move_s": float(oblk["move time"]), "oracle_accumulator_s": float(oblk["accumulator time"]),
                      "note": "same NumPy AO routine on both sides; single thread"}))


if __name__ == "__main__":
    main()
