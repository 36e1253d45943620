cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for lib in pyqmc_amd/lib/libpyqmc_amd.so pyqmc_amd/lib/ab/libpqa_tm3.so pyqmc_amd/lib/ab/libpqa_tm4.so; do
  for w in 4096 16384; do
  echo -n "$lib c5@$w "; PQA_LIB=$PWD/$lib timeout 120 python tools/config_bench.py c5 --walkers $w --steps 10 2>/dev/null | tail -1 | sed 's/.*ms_per_step/ms_per_step/' | cut -c1-70
  done
  rm -rf /tmp/pc; PQA_LIB=$PWD/$lib timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/pc -o k -- python tools/config_bench.py c5 --walkers 4096 --steps 10 > /tmp/pc.log 2>&1 < /dev/null
  python tools/prof_stats.py /tmp/pc/k_results.db | grep tm_walker | sed 's/(SysDev[^"]*"/"/' | cut -c1-100
done
