"""Phase stamps of k_step_pre's first 256 blocks, last launch of a sweep (build: python -c "import __graft_entry__ as g, os;
g.build(extra_flags=['-DPQA_PRE_CLK'], lib=os.path.join(g.LIBDIR, 'libpqa_PCLK.so'))"; run with PQA_LIB=pyqmc_amd/lib/libpqa_PCLK.so).
Stamps (thread 0 of a block): 0 entry, 1 every load issued, 2 Slater sums of the decide half done (= first two round trips waited for),
3 its Jastrow pairs done, 4 past the barrier, 5 totals (second barrier), 6 decided, 7 V / R in LDS (barrier), 8 block row committed
(barrier), 9 propose half's sums + barrier, 10 totals, 11 end (lead group: drift, proposal stored)."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pyqmc_amd as pa
from pyqmc_amd import _ffi
from tests import helpers
from pyqmc_amd import systems

W = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
mol = systems.water_cluster()
wf = helpers.gpu_wf(mol, systems.random_mf(mol))
dev = wf.fused_device()
wf.recompute(pa.initial_guess(mol, W, rng=np.random.default_rng(1)))
dev.vmc_sweeps(0.3, 3, seed=5, energy=False)
lib = _ffi.lib()
buf = (ctypes.c_ulonglong * (256 * 16))()
lib.pqa_debug_pre_clk.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert lib.pqa_debug_pre_clk(buf, 256 * 16) == 0
c = np.array(buf[:], dtype=np.float64).reshape(256, 16)[: min(256, W // 16)]
d = (c - c[:, :1]) / 100.0
names = ["entry", "loads issued", "slater sums (loads waited)", "jastrow pairs", "barrier 1", "totals + barrier 2", "decided", "V,R + barrier 3", "row commit + barrier 4",
         "propose sums + barrier 5", "totals + barrier 6", "end"]
print("blocks", len(c), "entry spread us %.2f" % ((c[:, 0].max() - c[:, 0].min()) / 100.0))
prev = np.zeros(len(c))
names += ["  commit: dot done", "  commit: row updated", "  commit: stores issued", ""]
for k in (1, 2, 3, 4, 5, 6, 7, 12, 13, 14, 8, 9, 10, 11):
    ok = c[:, k] > 0
    print("%-30s at %6.2f us (step %5.2f)   min %6.2f max %6.2f" % (names[k], d[ok, k].mean(), (d[ok, k] - prev[ok]).mean(), d[ok, k].min(), d[ok, k].max()))
    prev = np.where(ok, d[:, k], prev)
print("last block end since first entry: %.2f us" % ((c[:, 8].max() - c[:, 0].min()) / 100.0))
