cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for v in pyqmc_amd/lib/libpyqmc_amd.so pyqmc_amd/lib/variants/pre1.so pyqmc_amd/lib/variants/pre2.so pyqmc_amd/lib/variants/pre3.so pyqmc_amd/lib/variants/pre4.so; do
  rm -rf /tmp/pp; rocprofv3 --kernel-trace --stats -d /tmp/pp -o k -- python tools/scratch/r3_pre_abl.py $v 4096 > /tmp/pp.log 2>&1 < /dev/null
  echo -n "$v: "; python tools/prof_stats.py /tmp/pp/k_results.db | grep prepass | cut -c1-120
done
