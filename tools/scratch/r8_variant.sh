#!/bin/bash
# A/B variants of k_sweep_r8 without rebuilding the library: compile pqa_res8.hip alone (timing build, -DPQA_RES_CLK + the flags given) and
# link it with the other objects of pyqmc_amd/lib/obj_libpqa_RCLK (python -c "import __graft_entry__ as g, os;
# g.build(extra_flags=['-DPQA_RES_CLK'], lib=os.path.join(g.LIBDIR, 'libpqa_RCLK.so'))" first).
# usage: bash tools/scratch/r8_variant.sh NAME [-DFLAG ...]  ->  pyqmc_amd/lib/libpqa_NAME.so
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
name=$1; shift
OBJ=$ROOT/pyqmc_amd/lib/obj_libpqa_RCLK
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DPQA_RES_CLK -mllvm -disable-machine-licm "$@" -c $ROOT/pyqmc_amd/csrc/pqa_res8.hip -o /tmp/pqa_res8_$name.o 2>&1 | grep -E "error" -A5 || true
objs=$(ls $OBJ/*.o | grep -v pqa_res8.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/pqa_res8_$name.o -o $ROOT/pyqmc_amd/lib/libpqa_$name.so
echo built libpqa_$name.so
