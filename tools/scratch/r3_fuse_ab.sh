# round 3: fused k_step_lw (PQA_LW_FUSE=1) vs six-launch sequence (0): parity tests, bitwise comparison, timings
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3_fuse; mkdir -p $O
timeout 600 python tools/scratch/r3_fuse_check.py 4096 > $O/check.txt 2>&1
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt >> $O/check.txt
for f in 0 1; do for w in 1024 4096 16384 65536; do
  echo -n "FUSE=$f W=$w " >> $O/ab.txt
  PQA_LW_FUSE=$f timeout 300 python tools/scratch/lib_bench.py pyqmc_amd/lib/libpyqmc_amd.so $w >> $O/ab.txt 2>&1
done; done
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/p6 -o b -- python $GRAFT_REPO_ROOT/bench.py --walkers 65536 --steps 10 --warmup 2 --no-cpu-baseline --no-extra > /dev/null 2>&1 < /dev/null
python $GRAFT_REPO_ROOT/tools/prof_stats.py /tmp/p6/b_results.db $GRAFT_REPO_ROOT/$O/kernel_stats_65536.csv
rocprofv3 --kernel-trace --stats -d /tmp/p4 -o b -- python $GRAFT_REPO_ROOT/bench.py --walkers 4096 --steps 10 --warmup 2 --no-cpu-baseline --no-extra > /dev/null 2>&1 < /dev/null
python $GRAFT_REPO_ROOT/tools/prof_stats.py /tmp/p4/b_results.db $GRAFT_REPO_ROOT/$O/kernel_stats_4096.csv
cat $GRAFT_REPO_ROOT/$O/check.txt $GRAFT_REPO_ROOT/$O/ab.txt
