R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for v in BASE NOJAS NOSLATER; do
  lib=$R/tools/scratch/lib_$v.so; [ $v = BASE ] && lib=$R/pyqmc_amd/lib/libpyqmc_amd.so
  rocprofv3 --kernel-trace --stats -d /tmp/mp_$v -o t -- python $R/tools/scratch/lib_bench.py $lib > /dev/null 2>&1 < /dev/null
  echo "$v: $(python $R/tools/prof_stats.py /tmp/mp_$v/t_results.db | grep k_move_part | cut -c1-30,100-160)"
done
