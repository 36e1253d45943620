"""A/B of the pipelined half-ensemble sweep (PQA_SPLIT modes, pqa_sweep.hip) on one box: time per step and whether the
trajectory is bit-identical to the single-stream sweep.
usage: python tools/split_ab.py [--walkers W] [--steps K] [--system m|k222|c3] MODE[:CUS] ...   (mode 0 is always run first)"""
import argparse, hashlib, json, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

ap = argparse.ArgumentParser()
ap.add_argument("--walkers", type=int, default=65536)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--system", default="m")
ap.add_argument("--energy", type=int, default=1)
ap.add_argument("modes", nargs="*", default=["1", "2", "3"])
args = ap.parse_args()
import pyqmc_amd as pa
from pyqmc_amd import pbc


def build():
    if args.system == "m":
        import bench
        return bench.build_wf(0)[::2]
    if args.system == "k222":
        sup = pbc.get_supercell(pa.systems.diamond_primitive(), 2.0 * np.eye(3))
        return sup, pa.generate_wf(sup, pbc.random_kmf(sup), device=0)
    raise SystemExit("unknown system")


def run(mode):
    """mode: "<split mode>[:<orbital-stream CUs>][,KEY=VALUE ...]" (further PQA_* switches, e.g. 0,PQA_JPRE=1)"""
    head, *extra = mode.split(",")
    m, _, cus = head.partition(":")
    for k in [k for k in os.environ if k.startswith("PQA_")]:
        del os.environ[k]
    os.environ["PQA_SPLIT"] = m
    os.environ["PQA_SPLIT_CUS"] = cus or "0"
    os.environ["PQA_JPRE"] = "0"
    for kv in extra:
        k, _, v = kv.partition("=")
        os.environ[k] = v
    mol, wf = build()
    dev = wf.fused_device()
    wf.recompute(pa.initial_guess(mol, args.walkers, rng=np.random.default_rng(5)))
    dev.vmc_sweeps(0.3, 3, seed=1, energy=bool(args.energy))
    dev.sync()
    best, en = 1e9, None
    for rep in range(3):
        t0 = time.perf_counter()
        acc, en, _ = dev.vmc_sweeps(0.3, args.steps, seed=2 + rep, energy=bool(args.energy))
        dev.sync()
        best = min(best, (time.perf_counter() - t0) / args.steps)
    x = dev.configs()
    hsh = hashlib.sha256(np.ascontiguousarray(x).tobytes() + (en.tobytes() if en is not None else b"")).hexdigest()[:16]
    del wf, dev
    return {"mode": mode, "ms_per_step": round(1e3 * best, 3), "walker_steps_per_s": round(args.walkers / best), "hash": hsh,
            "acceptance": float(np.mean(acc))}


ref = run("0")
print(json.dumps(ref), flush=True)
for mode in args.modes:
    r = run(mode)
    r["bitwise_equal_to_mode0"] = r["hash"] == ref["hash"]
    print(json.dumps(r), flush=True)
