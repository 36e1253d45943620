# round 3 quick evaluation of the headline step: GPU parity suite, step times over W, kernel stats, HBM counters
# usage: bash tools/scratch/r3_eval.sh tag [nopmc]
cd $GRAFT_REPO_ROOT; T=${1:-eval}; O=$GRAFT_REPO_ROOT/gpurun_out/r3_$T; mkdir -p $O; rm -f $O/*.txt
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt > $O/summary.txt
for w in 1024 2048 4096 8192 16384 32768 65536; do
  echo -n "W=$w " >> $O/summary.txt; timeout 300 python tools/scratch/lib_bench.py pyqmc_amd/lib/libpyqmc_amd.so $w >> $O/summary.txt 2>&1
done
cd /tmp; export TMPDIR=/tmp
for w in 65536 4096; do
  rm -rf /tmp/pp; rocprofv3 --kernel-trace --stats -d /tmp/pp -o b -- python $GRAFT_REPO_ROOT/tools/scratch/lib_bench.py $GRAFT_REPO_ROOT/pyqmc_amd/lib/libpyqmc_amd.so $w > /tmp/pp.log 2>&1 < /dev/null
  python $GRAFT_REPO_ROOT/tools/prof_stats.py /tmp/pp/b_results.db $O/kernel_stats_$w.csv
done
if [ "$2" != "nopmc" ]; then
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pm_$c /tmp/pc_$c
  rocprofv3 --pmc $c -d /tmp/pm_$c -o t -- python $GRAFT_REPO_ROOT/bench.py --walkers 65536 --steps 2 --warmup 1 --settle 2 --no-cpu-baseline --no-profile --no-extra > /dev/null 2>&1 < /dev/null
  rocprofv3 --pmc $c -d /tmp/pc_$c -o t -- $GRAFT_REPO_ROOT/tools/pmc_calib > /dev/null 2>&1 < /dev/null
done
python $GRAFT_REPO_ROOT/tools/pmc_summary.py /tmp/pm_FETCH_SIZE/t_results.db /tmp/pm_WRITE_SIZE/t_results.db /tmp/pc_FETCH_SIZE/t_results.db /tmp/pc_WRITE_SIZE/t_results.db 65536 $O/pmc_summary.json > $O/pmc_summary.txt 2>&1
fi
cat $O/summary.txt; head -14 $O/kernel_stats_65536.csv; head -8 $O/kernel_stats_4096.csv; tail -25 $O/pmc_summary.txt
