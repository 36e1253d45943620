"""Where the protocol route's time goes (pyqmc.method.mc.vmc_worker unchanged over the pyqmc_amd objects).

    python tools/protocol_profile.py [walkers] [--json out.json]

Runs one protocol-route VMC step of the metric system ((H2O)8, 64 electrons) with every C-ABI call timed (wall clock around the
ctypes call: launch + synchronisation + copies it performs), next to the fused route.  The remainder is host NumPy / Python of the
reference's loop body (mc.py:115-137) and of the wrappers."""
import collections
import json
import os
import sys
import time

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import helpers  # noqa: E402
import pyqmc_amd as pa  # noqa: E402
from pyqmc_amd import wf as wfmod  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
W = int(args[0]) if args else 4096
out_path = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
mol = pa.systems.water_cluster()
wf = pa.generate_wf(mol, pa.systems.random_mf(mol))
acc = {"energy": pa.EnergyAccumulator(mol)}
calls = collections.defaultdict(lambda: [0, 0.0])
orig_call = wfmod.DeviceWF.call


def timed_call(self, name, *a):
    t0 = time.perf_counter()
    orig_call(self, name, *a)
    c = calls[name]
    c[0] += 1
    c[1] += time.perf_counter() - t0


res = {"walkers": W, "nelec": 64}
for route in ("fused", "protocol"):
    cfg = pa.initial_guess(mol, W, rng=np.random.default_rng(1))
    run = pa.vmc_worker if route == "fused" else helpers.protocol_vmc_worker
    run(wf, cfg, 0.3, 1, acc)  # warm-up
    n = 4 if route == "fused" else 2
    if route == "protocol":
        wfmod.DeviceWF.call = timed_call
    t0 = time.perf_counter()
    run(wf, cfg, 0.3, n, acc)
    dt = time.perf_counter() - t0
    wfmod.DeviceWF.call = orig_call
    res[route] = {"ms_per_step": 1e3 * dt / n, "walker_steps_per_s": W * n / dt}
    if route == "protocol":
        tot = sum(v[1] for v in calls.values())
        res[route]["c_abi_ms_per_step"] = 1e3 * tot / n
        res[route]["host_python_numpy_ms_per_step"] = 1e3 * (dt - tot) / n
        res[route]["calls"] = {k: {"per_step": v[0] / n, "ms_per_call": 1e3 * v[1] / v[0], "ms_per_step": 1e3 * v[1] / n}
                               for k, v in sorted(calls.items(), key=lambda kv: -kv[1][1])}
print(json.dumps(res, indent=1))
if out_path:
    json.dump(res, open(out_path, "w"), indent=1)
