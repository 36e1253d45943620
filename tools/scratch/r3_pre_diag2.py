"""debug build -DPQA_PRE_DBG=32: accept_rec holds mismatch codes of the proposal half's Jastrow sums (jas_pre vs jas_eval_lane)"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from pyqmc_amd import _ffi
_ffi.LIB_PATH = os.path.abspath(sys.argv[1])
import pyqmc_amd as pa
W = 4096
os.environ["PQA_STEP_PRE"] = "1"
mol = pa.systems.water_cluster(); mf = pa.systems.random_mf(mol)
wf = pa.generate_wf(mol, mf); dev = wf.fused_device()
wf.recompute(pa.initial_guess(mol, W, rng=np.random.default_rng(11)))
acc = np.empty(1); rec = np.zeros((1, dev.N, W), dtype=np.uint8)
dev.call("pqa_vmc_sweeps", 0.3, 1, _ffi.ptr(None), _ffi.ptr(None), 10.0, _ffi.ptr(None), _ffi.ptr(None), 21, _ffi.ptr(acc), _ffi.ptr(None), _ffi.ptr(rec))
r = rec[0]
print("mismatching (electron, walker) pairs:", int((r != 0).sum()), "of", r.size)
codes, counts = np.unique(r & 15, return_counts=True); print("component codes (1 U, 2 gx, 4 gy, 8 gz):", dict(zip(codes.tolist(), counts.tolist())))
grp, gc = np.unique((r[r != 0] >> 4), return_counts=True); print("groups:", dict(zip(grp.tolist(), gc.tolist())))
el, ec = np.unique(np.argwhere(r != 0)[:, 0], return_counts=True); print("electrons:", dict(zip(el.tolist(), ec.tolist())))
