"""Phase stamps of k_orb_wide's first blocks (build: python -c "import __graft_entry__ as g, os; g.build(extra_flags=['-DPQA_WIDE_CLK'], lib=os.path.join(g.LIBDIR, 'libpqa_CLK.so'))";
run with PQA_LIB=pyqmc_amd/lib/libpqa_CLK.so).  Stamps: 0 entry, 1 tables staged + tile zeroed, 2 wave 0 done with phase 1 (6: last wave),
3 past the barrier, 4 wave 0's last role contracted, 5 end."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pyqmc_amd as pa
from pyqmc_amd import pbc, _ffi

W = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
sup = pbc.get_supercell(pa.systems.diamond_primitive(), 2.0 * np.eye(3))
mf = pbc.random_kmf(sup)
dev = pa.generate_wf(sup, mf).fused_device()
pts = np.random.default_rng(1).random((W, 3)) @ sup.lattice_vectors()
for _ in range(3):
    dev.eval_mo(0, pts, 5)
lib = _ffi.lib()
buf = (ctypes.c_ulonglong * (1024 * 8))()
lib.pqa_debug_wide_clk.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert lib.pqa_debug_wide_clk(buf, 1024 * 8) == 0
c = np.array(buf[:], dtype=np.float64).reshape(1024, 8)[: min(1024, W // 16)]
t0 = c[:, 0].min()
d = (c - c[:, :1]) / 100.0  # us since the block's own entry
print("blocks", len(c), "entry spread us", (c[:, 0].max() - t0) / 100.0)
for k, name in [(1, "staged+zeroed"), (2, "wave0 phase1 done"), (6, "last wave phase1 done"), (3, "past barrier"), (4, "wave0 contracted"), (5, "end")]:
    print("%-22s mean %6.2f  min %6.2f  max %6.2f us" % (name, d[:, k].mean(), d[:, k].min(), d[:, k].max()))
print("last block end since first entry: %.2f us" % ((c[:, 5].max() - t0) / 100.0))
