"""Resident sweep (PQA_RES=1) against the launch-per-move sweep (PQA_RES=0): decisions, state, timing.
usage: python tools/scratch/res_check.py [check|time] ..."""
import os, sys, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np

def make(res, W, system="M"):
    os.environ["PQA_RES"] = str(res)
    import pyqmc_amd as pa
    mol = pa.systems.water_cluster() if system == "M" else pa.systems.water()
    mf = pa.systems.random_mf(mol)
    wf = pa.generate_wf(mol, mf)
    dev = wf.fused_device()
    wf.recompute(pa.initial_guess(mol, W, rng=np.random.default_rng(11)))
    return mol, wf, dev

def check(W, system):
    outs = []
    for res in (0, 1):
        mol, wf, dev = make(res, W, system)
        acc, en, rec = dev.vmc_sweeps(0.3, 2, seed=21, energy=True, record=True)
        x = dev.configs(); logv = dev.value()[1]
        rv = dev.recompute(x)[1]
        outs.append(dict(rec=rec, acc=np.asarray(acc), x=x, logv=logv, en=np.asarray(en), upd=np.max(np.abs(rv - logv))))
    a, b = outs
    same = np.array_equal(a["rec"], b["rec"])
    ndiff = int(np.sum(a["rec"] != b["rec"]))
    print(json.dumps(dict(system=system, W=W, decisions_equal=bool(same), ndiff=ndiff, acc=[float(a["acc"].mean()), float(b["acc"].mean())],
                          dx=float(np.max(np.abs(a["x"] - b["x"]))), dlogv=float(np.max(np.abs(a["logv"] - b["logv"]))),
                          den=float(np.max(np.abs(a["en"] - b["en"]))), upd_vs_recompute=[float(a["upd"]), float(b["upd"])])), flush=True)

def timeit(W, res, system="M", energy=True, nst=4):
    mol, wf, dev = make(res, W, system)
    dev.vmc_sweeps(0.3, 2, seed=1, energy=energy); dev.sync()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter(); dev.vmc_sweeps(0.3, nst, seed=2 + rep, energy=energy); dev.sync()
        best = min(best, (time.perf_counter() - t0) / nst)
    print(json.dumps(dict(system=system, W=W, res=res, energy=energy, ms_per_step=round(1e3 * best, 3), walker_steps_per_s=round(W / best))), flush=True)

if __name__ == "__main__":
    mode = sys.argv[1]
    if mode == "check":
        for system, W in (("M", 1000), ("M", 4096), ("C2", 530)):
            check(W, system)
    else:
        for W in [int(x) for x in sys.argv[2:]] or [4096, 65536]:
            for res in (0, 1):
                for energy in (False, True):
                    timeit(W, res, energy=energy)
