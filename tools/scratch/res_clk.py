"""Phase stamps of k_sweep_res's first 64 blocks, last move of a sweep (build: python -c "import __graft_entry__ as g, os;
g.build(extra_flags=['-DPQA_RES_CLK'], lib=os.path.join(g.LIBDIR, 'libpqa_RCLK.so'))"; run with PQA_LIB=pyqmc_amd/lib/libpqa_RCLK.so).
Stamps (thread 0 of a block, 100 MHz): 0 move entry (before the barrier), 1 AO phase done, 2 contraction done, 3 partials combined-ready
(two barriers), 4 decided (Slater sums, Jastrow, Metropolis), 5 committed (Sherman-Morrison, cache row), 6 next electron proposed
(stamp 6 is from the move before the last)."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pyqmc_amd as pa
from pyqmc_amd import _ffi, systems
from tests import helpers

W = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
mol = systems.water() if len(sys.argv) > 2 and sys.argv[2] == "c2" else systems.water_cluster()
wf = helpers.gpu_wf(mol, systems.random_mf(mol))
dev = wf.fused_device()
wf.recompute(pa.initial_guess(mol, W, rng=np.random.default_rng(1)))
dev.vmc_sweeps(0.3, 3, seed=5, energy=False)
lib = _ffi.lib()
buf = (ctypes.c_ulonglong * (64 * 16))()
r8 = os.environ.get("PQA_R8", "-1") != "0" and len(sys.argv) <= 2  # (the 64-electron cluster runs k_sweep_r8 unless PQA_R8=0)
fn = lib.pqa_debug_r8_clk if r8 else lib.pqa_debug_res_clk
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert fn(buf, 64 * 16) == 0
c = np.array(buf[:], dtype=np.float64).reshape(64, 16)[: min(64, (W + 15) // 16)]
print("kernel", "k_sweep_r8" if r8 else "k_sweep_res")
seq = [(0, "move entry"), (1, "AO phase (+ barrier wait before)"), (2, "contraction (+ barrier)"), (3, "partials + 2 barriers"), (7, "rows combined"),
       (8, "Slater sums (4 x sum32)"), (9, "Jastrow at the proposal (+ 4 x sum32)"), (4, "Metropolis"), (13, "Sherman-Morrison (accepted)"), (5, "cache row, selector")]
if r8:  # k_sweep_r8: both Jastrow evaluations run ahead of the orbitals (stamp 9 before the AO phase's barrier)
    seq = [(0, "move entry"), (9, "Jastrow, both evaluations (+ 12 x sum32)"), (1, "barrier + AO phase"), (2, "contraction (+ barrier)"), (3, "partials + 2 barriers"),
           (7, "rows combined"), (8, "Slater sums (4 x sum32)"), (4, "Metropolis"), (13, "Sherman-Morrison (accepted)"), (5, "cache row, selector")]
print("walkers", W, "blocks sampled", len(c))
prev = 0
for k, name in seq[1:]:
    ref = c[:, prev]
    if k == 5:  # stamp 13 (Sherman-Morrison) is written by accepted moves only: a rejected move goes from the Metropolis stamp to stamp 5
        ref = np.where(c[:, 13] >= c[:, 4], c[:, 13], c[:, 4])
    ok = c[:, k] >= ref
    d = (c[ok, k] - ref[ok]) / 100.0
    if len(d): print("%-36s %6.2f us  (min %5.2f max %5.2f, n %d)" % (name, d.mean(), d.min(), d.max(), ok.sum()))
    prev = k
print("%-36s %6.2f us" % ("entry -> committed", ((c[:, 5] - c[:, 0]) / 100.0).mean()))
print("previous move's proposal:")
prev = 5
for k, name in [(10, "rowE handed over"), (11, "Slater sums"), (12, "Jastrow at the current position"), (6, "drift, proposal")]:
    d = (c[:, k] - c[:, 10 if k != 10 else k]) / 100.0
    print("%-36s at %6.2f us after the hand-over" % (name, d.mean()))

if r8:
    d = (c[:, 15] - c[:, 14]) / 100.0
    print("block lifetime (entry -> end of the sweep): mean %.1f us, min %.1f, max %.1f; per move %.2f us" % (d.mean(), d.min(), d.max(), d.mean() / 64))
    fn2 = lib.pqa_debug_r8_clk2
    fn2.argtypes = [ctypes.c_void_p, ctypes.c_int]
    buf2 = (ctypes.c_ulonglong * (64 * 16))()
    assert fn2(buf2, 64 * 16) == 0
    c2 = np.array(buf2[:], dtype=np.float64).reshape(64, 16)[: len(c)]
    t0 = c[:, 14]
    names = ["tables in LDS", "spin 0: rows loaded", "spin 0: moves done", "spin 0: state stored", "spin 0: closing barrier",
             "spin 1: rows loaded", "spin 1: moves done", "spin 1: state stored", "spin 1: closing barrier"]
    prev = t0
    for k, nm in enumerate(names):
        d = (c2[:, k] - prev) / 100.0
        print("  %-26s +%7.2f us (min %6.2f max %7.2f)" % (nm, d.mean(), d.min(), d.max()))
        prev = c2[:, k]
    print("  %-26s +%7.2f us" % ("end of the sweep", ((c[:, 15] - prev) / 100.0).mean()))
