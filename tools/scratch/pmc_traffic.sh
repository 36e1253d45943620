# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of the bench's kernels -> gpurun_out/pmc_fetch.csv, pmc_write.csv
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d /tmp/pm_$c -o t -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile > /dev/null 2>&1 < /dev/null
  python $R/tools/pmc_counters.py /tmp/pm_$c/t_results.db $R/gpurun_out/pmc_$c.csv
  grep "k_flush_lw\|k_commit_lw\|k_orbILi5\|k_orb<5\|k_move_part" $R/gpurun_out/pmc_$c.csv | cut -c1-60,100-220
done
