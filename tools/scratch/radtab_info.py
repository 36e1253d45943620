"""radial_table_info (table doubles, fit error) of a few systems, incl. tight all-electron primitives."""
import os, sys, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import pyqmc_amd as pa
from pyqmc_amd import _ffi, systems
for name, mol in (("water ccECP", systems.water()), ("cluster", systems.water_cluster()), ("water all-electron general contractions", systems.water_general())):
    mf = systems.random_mf(mol)
    dev = pa.DeviceWF(mol, mo_coeff=mf.mo_coeff)
    info = np.zeros(2)
    assert _ffi.lib().pqa_get_param(dev._h, b"radial_table_info", info.ctypes.data_as(C.c_void_p), 2) == 0
    exps = np.concatenate([np.asarray(b[1:])[:, 0] for a in mol._basis.values() for b in a]) if hasattr(mol, "_basis") else None
    print(name, "table doubles", info[0], "fit error", info[1], "largest exponent", None if exps is None else exps.max())
