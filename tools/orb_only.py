"""Run only the orbital kernel (both schedules) on the metric system: for rocprofv3 PMC passes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pyqmc_amd as pa
npts = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
mol = pa.systems.water_cluster(); mf = pa.systems.random_mf(mol)
dev = pa.DeviceWF(mol, mo_coeff=mf.mo_coeff)
rng = np.random.default_rng(0)
pts = mol.atom_coords()[rng.integers(mol.natm, size=npts)] + rng.standard_normal((npts, 3))
for rep in range(3):
    dev.eval_mo(0, pts, 5, use_mfma=True)
print("done")
