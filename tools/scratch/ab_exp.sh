cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests -x -q -m gpu 2>&1 | tail -2
bash tools/scratch/ab_libs.sh "k_orb" pyqmc_amd/lib/libpyqmc_amd.so pyqmc_amd/lib/ab/libpqa_libexp.so
for lib in pyqmc_amd/lib/libpyqmc_amd.so pyqmc_amd/lib/ab/libpqa_libexp.so; do
  echo -n "$lib c5@4096 "; PQA_LIB=$PWD/$lib timeout 120 python tools/config_bench.py c5 --walkers 4096 --steps 10 2>/dev/null | tail -1 | sed 's/.*ms_per_step/ms_per_step/' | cut -c1-70
  echo -n "$lib k222@32768 "; PQA_LIB=$PWD/$lib timeout 120 python tools/pbc_bench.py --case k222 --walkers 32768 --steps 3 2>/dev/null | tail -1 | sed 's/.*ms_per_step/ms_per_step/' | cut -c1-70
done
