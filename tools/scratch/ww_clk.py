"""Phase stamps inside k_propose / k_accept (wave-per-walker move kernels, BASELINE config C4).  Build:
python -c "import __graft_entry__ as g, os; g.build(extra_flags=['-DPQA_WW_CLK'], lib=os.path.join(g.LIBDIR, 'libpqa_WCLK.so'))"; run with
PQA_LIB=pyqmc_amd/lib/libpqa_WCLK.so.  Stamps: k_propose 0 entry, 1 Slater terms, 2 two-body Jastrow, 3 three-body; k_accept 4 entry, 5 Slater terms,
6 Jastrow + decision, 7 end (accepted walkers: Sherman-Morrison on every determinant)."""
import ctypes, os, subprocess, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.argv = [sys.argv[0], "c4", "--walkers", "2048", "--steps", "2"]
_cb = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "config_bench.py")
exec(compile(open(_cb).read(), _cb, "exec"), {"__file__": _cb, "__name__": "__main__"})
from pyqmc_amd import _ffi
lib = _ffi.lib()
buf = (ctypes.c_ulonglong * (1024 * 8))()
lib.pqa_debug_ww_clk.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert lib.pqa_debug_ww_clk(buf, 1024 * 8) == 0
c = np.array(buf[:], dtype=np.float64).reshape(1024, 8)
for base, names in [(0, ["Slater terms", "two-body Jastrow", "three-body Jastrow"]), (4, ["Slater terms", "Jastrow + decision", "commit (accepted)"])]:
    print("k_propose" if base == 0 else "k_accept")
    for k, nm in enumerate(names):
        d = (c[:, base + k + 1] - c[:, base + k]) / 100.0
        d = d[(d >= 0) & (d < 1000)]
        print("  %-22s mean %6.2f  median %6.2f  max %6.2f us" % (nm, d.mean(), np.median(d), d.max()))
