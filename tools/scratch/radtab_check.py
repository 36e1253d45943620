"""Per-walker energies and walkers of the headline system with and without the radial tables (PQA_RADTAB), same Philox streams."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
out = {}
W = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
for t in ("0", "1", "0"):
    os.environ["PQA_RADTAB"] = t
    import pyqmc_amd as pa
    mol = pa.systems.water_cluster(); mf = pa.systems.random_mf(mol)
    wf = pa.generate_wf(mol, mf); dev = wf.fused_device()
    wf.recompute(pa.initial_guess(mol, W, rng=np.random.default_rng(11)))
    acc, en, rec = dev.vmc_sweeps(0.3, 3, seed=21, energy=True, record=True)
    e = dev.energy(10.0, seed=9)
    out.setdefault(t, []).append((np.asarray(rec), dev.configs(), np.asarray(e)))
a, b, a2 = out["0"][0], out["1"][0], out["0"][1]
print("repro radtab0: decisions", np.array_equal(a[0], a2[0]), "x", np.abs(a[1] - a2[1]).max(), "energy rows", np.abs(a[2] - a2[2]).max())
print("radtab 1 vs 0: decisions equal", np.array_equal(a[0], b[0]), "flips", int((a[0] != b[0]).sum()), "x", np.abs(a[1] - b[1]).max())
d = np.abs(a[2] - b[2])
print("energy rows max abs diff per row", d.max(axis=1), "mean diff of total", float(np.mean(a[2][5] - b[2][5])))
