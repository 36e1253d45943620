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
from pyqmc_amd import pbc as _pbc
if len(sys.argv) > 2 and sys.argv[2] == "c3":  # twisted conventional cell, complex orbitals
    mol = _pbc.get_supercell(systems.diamond_primitive(), np.array([[-1.0, 1, 1], [1, -1, 1], [1, 1, -1]]))
    wf = pa.generate_wf(mol, _pbc.random_kmf(mol, complex_coeff=True, twist=(0.25, 0.1, -0.3)))
else:
    mol = _pbc.get_supercell(systems.diamond_primitive(), 2.0 * np.eye(3))
    wf = pa.generate_wf(mol, _pbc.random_kmf(mol))
dev = wf.fused_device()
wf.recompute(pa.initial_guess(mol, W, rng=np.random.default_rng(1)))
dev.vmc_sweeps(0.3, 3, seed=5, energy=False)
lib = _ffi.lib()
buf = (ctypes.c_ulonglong * (64 * 16))()
lib.pqa_debug_res_clk.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert lib.pqa_debug_res_clk(buf, 64 * 16) == 0
c = np.array(buf[:], dtype=np.float64).reshape(64, 16)[: min(64, (W + 15) // 16)]
seq = [(0, "move entry"), (1, "AO phase (+ barrier wait before)"), (2, "contraction (+ barrier)"), (3, "partials + 2 barriers"), (7, "rows combined"),
       (8, "Slater sums (4 x sum32)"), (9, "Jastrow at the proposal (+ 4 x sum32)"), (4, "Metropolis"), (13, "Sherman-Morrison (accepted)"), (5, "cache row, selector")]
print("walkers", W, "blocks sampled", len(c))
raw = np.array(buf[:], dtype=np.uint64).reshape(64, 16)[: len(c)]
print("phase 0 (image lists) us", ((c[:, 14] - c[:, 0]) / 100.0).mean(), " pairs flagged 255:", (raw[:, 15] >> np.uint64(32)).mean(), " list entries per block:", (raw[:, 15] & np.uint64(0xffffffff)).mean())
prev = 0
for k, name in seq[1:]:
    ok = c[:, k] >= c[:, prev]
    d = (c[ok, k] - c[ok, prev]) / 100.0
    print("%-36s %6.2f us  (min %5.2f max %5.2f, n %d)" % (name, d.mean(), d.min(), d.max(), ok.sum()))
    prev = k
print("%-36s %6.2f us" % ("entry -> committed", ((c[:, 5] - c[:, 0]) / 100.0).mean()))
print("previous move's proposal:")
prev = 5
for k, name in [(10, "rowE handed over"), (11, "Slater sums"), (12, "Jastrow at the current position"), (6, "drift, proposal")]:
    d = (c[:, k] - c[:, 10 if k != 10 else k]) / 100.0
    print("%-36s at %6.2f us after the hand-over" % (name, d.mean()))

b2 = (ctypes.c_ulonglong * (64 * 16))()
lib.pqa_debug_res_clk2.argtypes = [ctypes.c_void_p, ctypes.c_int]
if lib.pqa_debug_res_clk2(b2, 64 * 16) == 0:
    q = np.array(b2[:], dtype=np.float64).reshape(64, 16)[: len(c)]
    print("thread 0's phase 0 (pairs dealt to several threads), shader cycles: fold + masks %.0f, first walk %.0f, barrier %.0f, offsets + second walk %.0f" % tuple(q[:, k].mean() for k in range(5, 9)))
    print("thread 0's phase 1, shader cycles (mean over blocks): header+zeroing %.0f, fold %.0f, walk+evaluation %.0f; shells %.1f, images evaluated %.1f" % tuple(q[:, k].mean() for k in range(5)))

b3 = (ctypes.c_ulonglong * (64 * 8))()
lib.pqa_debug_res_clk3.argtypes = [ctypes.c_void_p, ctypes.c_int]
if lib.pqa_debug_res_clk3(b3, 64 * 8) == 0:
    q = np.array(b3[:], dtype=np.float64).reshape(64, 8)[: len(c)]
    print("phase 1 per wave (lane 0 of waves 0..7), k cycles:", (q.mean(axis=0) / 1e3).round(1))
