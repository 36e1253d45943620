"""Where do k_step_pre and k_step_lw part ways?  python r3_pre_diag.py [W]"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import pyqmc_amd as pa
W = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
if len(sys.argv) > 2:
    from pyqmc_amd import _ffi
    _ffi.LIB_PATH = os.path.abspath(sys.argv[2])
outs = []
for pre in ("0", "1"):
    os.environ["PQA_STEP_PRE"] = pre
    mol = pa.systems.water_cluster(); mf = pa.systems.random_mf(mol)
    wf = pa.generate_wf(mol, mf); dev = wf.fused_device()
    wf.recompute(pa.initial_guess(mol, W, rng=np.random.default_rng(11)))
    acc, en, rec = dev.vmc_sweeps(0.3, 1, seed=21, energy=True, record=True)
    outs.append((dev.configs(), dev.value()[1], np.asarray(en), np.asarray(rec)))
a, b = outs
dx = np.abs(a[0] - b[0])
print("max |dx|", dx.max(), "walkers differing", int((dx.reshape(W, -1).max(axis=1) > 0).sum()), "of", W)
rec_a, rec_b = a[3].reshape(-1, W) if a[3].ndim > 2 else a[3], b[3].reshape(-1, W) if b[3].ndim > 2 else b[3]
print("rec shape", a[3].shape, "decisions differing", int((a[3] != b[3]).sum()))
de = dx.max(axis=2)  # (W, N)
bad = np.argwhere(de > 0)
print("first differing (walker, electron):", bad[:10].tolist())
if len(bad):
    w, e = bad[0]
    print("walker", w, "electron", e, "a", a[0][w, e], "b", b[0][w, e], "diff", a[0][w, e] - b[0][w, e])
    es = sorted(set(bad[:, 1].tolist()))
    print("electrons that differ anywhere:", es[:70])
print("dlog max diff", np.abs(a[1] - b[1]).max(), "energy diff", np.abs(a[2] - b[2]).max())
