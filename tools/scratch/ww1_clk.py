"""Phase stamps inside k_sweep_ww (the wave-per-walker sweep in one launch, BASELINE config C4), last move of the last sweep.  Build:
python -c "import __graft_entry__ as g, os; g.build(extra_flags=['-DPQA_WW_CLK'], lib=os.path.join(g.LIBDIR, 'libpqa_WCLK.so'))"; run with
PQA_LIB=pyqmc_amd/lib/libpqa_WCLK.so PQA_WW=1.  Per wave (0 Slater, 1 two-body Jastrow, 2 three-body): 0 phase A entry, 1 its part done, 2 past the
barrier, 3 phase C entry (proposal made), 4 its part done (wave 0: 7 = orbital row done), 5 past the barrier, 6 decided + committed."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
W = sys.argv[1] if len(sys.argv) > 1 else "1024"
sys.argv = [sys.argv[0], "c4", "--walkers", W, "--steps", "2"]
_cb = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "config_bench.py")
exec(compile(open(_cb).read(), _cb, "exec"), {"__file__": _cb, "__name__": "__main__"})
from pyqmc_amd import _ffi
lib = _ffi.lib()
buf = (ctypes.c_ulonglong * (256 * 3 * 8))()
lib.pqa_debug_ww1_clk.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert lib.pqa_debug_ww1_clk(buf, 256 * 3 * 8) == 0
c = np.array(buf[:], dtype=np.float64).reshape(256, 3, 8) / 100.0
t0 = c[:, :, 0].min(axis=1)[:, None, None]
c = c - t0
names = ["A entry", "A part done", "A barrier passed", "C entry", "C part done", "C barrier passed", "committed", "orbital row done (wave 0)"]
for wv in range(3):
    print("wave", wv, " ".join("%s %.2f" % (names[k], c[:, wv, k].mean()) for k in ([0, 1, 2, 3, 7, 4, 5, 6] if wv == 0 else range(7))))
