# round 3: two-slot row cache (no cache refresh) vs the fused build with the walker-fastest cache (variants/lib_base.so)
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r3_rc; mkdir -p $O; rm -f $O/*.txt
for w in 4096 20000; do
  timeout 300 python tools/scratch/r3_traj.py pyqmc_amd/lib/variants/lib_base.so /tmp/a_$w.npz $w >> $O/check.txt 2>&1
  timeout 300 python tools/scratch/r3_traj.py pyqmc_amd/lib/libpyqmc_amd.so /tmp/b_$w.npz $w >> $O/check.txt 2>&1
  python tools/scratch/r3_traj.py --cmp /tmp/a_$w.npz /tmp/b_$w.npz >> $O/check.txt 2>&1
done
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -15 $O/pytest.txt >> $O/check.txt
for w in 1024 2048 4096 8192 16384 32768 65536; do
  echo -n "base W=$w " >> $O/ab.txt; timeout 300 python tools/scratch/lib_bench.py pyqmc_amd/lib/variants/lib_base.so $w >> $O/ab.txt 2>&1
  echo -n "rc   W=$w " >> $O/ab.txt; timeout 300 python tools/scratch/lib_bench.py pyqmc_amd/lib/libpyqmc_amd.so $w >> $O/ab.txt 2>&1
done
cd /tmp; export TMPDIR=/tmp
for w in 65536 4096; do
  rm -rf /tmp/pp; rocprofv3 --kernel-trace --stats -d /tmp/pp -o b -- python $GRAFT_REPO_ROOT/tools/scratch/lib_bench.py $GRAFT_REPO_ROOT/pyqmc_amd/lib/libpyqmc_amd.so $w > /tmp/pp.log 2>&1 < /dev/null
  python $GRAFT_REPO_ROOT/tools/prof_stats.py /tmp/pp/b_results.db $O/kernel_stats_$w.csv
done
cat $O/check.txt $O/ab.txt; head -12 $O/kernel_stats_65536.csv
