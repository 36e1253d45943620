# per-variant step time + kernel averages at 65536 walkers: r3_var_stats.sh lib1.so lib2.so ...
cd /tmp; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r3_var; mkdir -p $O; rm -f $O/stats.txt
for v in "$@"; do
  rm -rf /tmp/pp; rocprofv3 --kernel-trace --stats -d /tmp/pp -o b -- python $GRAFT_REPO_ROOT/tools/scratch/lib_bench.py $GRAFT_REPO_ROOT/pyqmc_amd/lib/variants/$v 65536 > /tmp/pp.log 2>&1 < /dev/null
  python $GRAFT_REPO_ROOT/tools/prof_stats.py /tmp/pp/b_results.db /tmp/pp/s.csv
  echo "== $v $(grep 'ms/step' /tmp/pp.log)" >> $O/stats.txt
  python - <<'PY' >> $O/stats.txt
import csv
for r in csv.DictReader(open('/tmp/pp/s.csv')):
    n=r['kernel']
    if any(k in n for k in ('k_step_lw','k_orb','k_flush','k_kinetic_lw','k_ecp')): print('   ', n[:44].ljust(44), r['calls'], r['avg_us'], r['pct'])
PY
done
cd $GRAFT_REPO_ROOT; for v in "$@"; do echo -n "$v (no profiler) " >> $O/stats.txt; timeout 300 python tools/scratch/lib_bench.py pyqmc_amd/lib/variants/$v 65536 >> $O/stats.txt 2>&1; done
cat $O/stats.txt
