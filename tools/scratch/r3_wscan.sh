# round 3: small-shard scan of the lane-per-walker (PQA_LW=1) and walker-tile (PQA_LW=2) sweeps + kernel stats at 4096 walkers
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3_wscan; mkdir -p $O
for lw in 1 2; do for w in 1024 2048 4096 8192 16384; do
  echo -n "PQA_LW=$lw W=$w " >> $O/wscan.txt
  PQA_LW=$lw timeout 300 python tools/scratch/lib_bench.py pyqmc_amd/lib/libpyqmc_amd.so $w >> $O/wscan.txt 2>&1
  echo -n "PQA_LW=$lw W=$w sweep-only " >> $O/wscan.txt
  PQA_LW=$lw timeout 300 python tools/scratch/tile_time.py pyqmc_amd/lib/libpyqmc_amd.so $w >> $O/wscan.txt 2>&1
done; done
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/p4 -o b -- python $GRAFT_REPO_ROOT/bench.py --walkers 4096 --steps 10 --warmup 2 --no-cpu-baseline --no-extra > /dev/null 2>&1 < /dev/null
python $GRAFT_REPO_ROOT/tools/prof_stats.py /tmp/p4/b_results.db $GRAFT_REPO_ROOT/$O/kernel_stats_4096.csv
cat $GRAFT_REPO_ROOT/$O/wscan.txt
