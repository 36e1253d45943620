R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for v in BASE NOCACHE NOROWS; do
  lib=$R/tools/scratch/lib_$v.so; [ $v = BASE ] && lib=$R/pyqmc_amd/lib/libpyqmc_amd.so
  rocprofv3 --kernel-trace --stats -d /tmp/cq_$v -o t -- python $R/tools/scratch/lib_bench.py $lib > /dev/null 2>&1 < /dev/null
  echo "$v: $(python $R/tools/prof_stats.py /tmp/cq_$v/t_results.db | grep 'k_commit_lw' | cut -c1-30,110-170)"
done
