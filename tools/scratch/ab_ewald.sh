# k_ewald marking pass A/B: PQA_LIB = previous build vs HEAD
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for lib in pyqmc_amd/lib/libpqa_head.so pyqmc_amd/lib/libpyqmc_amd.so; do
  echo $lib
  PQA_LIB=$PWD/$lib python tools/pbc_bench.py --case k222 --walkers 32768 --steps 4 2>/dev/null | tail -1 | cut -c1-140
  d=/tmp/abe_$(basename $lib .so); rm -rf $d
  PQA_LIB=$PWD/$lib rocprofv3 --kernel-trace --stats -d $d -o r -- python tools/pbc_bench.py --case k222 --walkers 32768 --steps 3 > /dev/null 2>&1
  python tools/prof_stats.py $d/r_results.db 2>/dev/null | grep -E "k_ewald" | cut -c1-40,80-
done
