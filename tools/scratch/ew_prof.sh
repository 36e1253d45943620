R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
cat > /tmp/ew.py <<'PY'
import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from pyqmc_amd import _ffi
_ffi.LIB_PATH = os.path.abspath(sys.argv[1])
import numpy as np, pyqmc_amd as pa
from pyqmc_amd import pbc
sup = pbc.get_supercell(pa.systems.diamond_primitive(), 2.0 * np.eye(3))
wf = pa.generate_wf(sup, pbc.random_kmf(sup)); dev = wf.fused_device()
wf.recompute(pa.initial_guess(sup, 32768, rng=np.random.default_rng(1)))
dev.vmc_sweeps(0.3, 2, seed=1, energy=True); dev.sync()
PY
for v in BASE NOREAL NORECIP; do
  lib=$R/tools/scratch/lib_$v.so; [ $v = BASE ] && lib=$R/pyqmc_amd/lib/libpyqmc_amd.so
  rocprofv3 --kernel-trace --stats -d /tmp/ew_$v -o t -- python /tmp/ew.py $lib > /dev/null 2>&1 < /dev/null
  echo "$v: $(python $R/tools/prof_stats.py /tmp/ew_$v/t_results.db | grep k_ewald | cut -c1-20,75-130)"
done
