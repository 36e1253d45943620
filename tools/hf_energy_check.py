import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import pyqmc_amd as pa
from pyqmc_amd import chkfile, pbc
cell, mf = chkfile.load_scf("tests/golden/files/diamond_primitive.hdf5")
sup = pbc.get_supercell(cell, 2.0 * np.eye(3))
sl = pa.Slater(sup, mf)
dev = sl._dev
print("cplx", dev.cplx, "twisted", getattr(dev, "twisted", None), "N", dev.N)
W = 4096
cfg = pa.initial_guess(sup, W, rng=np.random.default_rng(1))
acc = {"energy": pa.EnergyAccumulator(sup)}
np.random.seed(3)
df, cfg = pa.vmc(sl, cfg, nblocks=12, nsteps_per_block=10, tstep=0.3, accumulators=acc, seed=5)
e = np.real(df["energytotal"])
print("blocks", e)
print("E_VMC (last 8 blocks)", e[4:].mean(), "+-", e[4:].std() / np.sqrt(len(e[4:])), " 8 x e_tot", 8 * mf.e_tot)
for k in ("energyke", "energyee", "energyei", "energyecp", "energyii"):
    if k in df: print(k, np.real(df[k])[4:].mean())
