cd /tmp; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r3_abl; mkdir -p $O; rm -f $O/abl2.txt
for v in base norefresh nocommit nojas noslater; do for w in 65536 4096; do
  rm -rf /tmp/pp; rocprofv3 --kernel-trace --stats -d /tmp/pp -o b -- python $GRAFT_REPO_ROOT/tools/scratch/lib_bench.py $GRAFT_REPO_ROOT/pyqmc_amd/lib/variants/lib_$v.so $w > /tmp/pp.log 2>&1 < /dev/null
  python $GRAFT_REPO_ROOT/tools/prof_stats.py /tmp/pp/b_results.db /tmp/pp/s.csv
  echo "== $v W=$w $(grep 'ms/step' /tmp/pp.log)" >> $O/abl2.txt
  python - <<'PY' >> $O/abl2.txt
import csv
for r in csv.DictReader(open('/tmp/pp/s.csv')):
    n=r['kernel']
    if any(k in n for k in ('k_step_lw','k_orb','k_flush','k_kinetic_lw')): print('   ', n[:40].ljust(40), r['calls'], r['avg_us'], r['pct'])
PY
done; done
cd $GRAFT_REPO_ROOT
for w in 8192 16384 32768; do for gm in 4 8 16; do
  echo -n "GM=$gm W=$w " >> $O/abl2.txt; PQA_LW_GM=$gm timeout 300 python tools/scratch/lib_bench.py pyqmc_amd/lib/libpyqmc_amd.so $w >> $O/abl2.txt 2>&1
done; done
cat $O/abl2.txt
