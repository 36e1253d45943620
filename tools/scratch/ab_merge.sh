R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ab_merge; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for m in 1 0; do
  rm -rf /tmp/pb$m; PQA_JAS_MERGE=$m rocprofv3 --kernel-trace --stats -d /tmp/pb$m -o b -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra > $O/bench_prof_m$m.json 2>/dev/null < /dev/null
  python $R/tools/prof_stats.py /tmp/pb$m/b_results.db $O/kernel_stats_m$m.csv
  PQA_JAS_MERGE=$m python $R/bench.py --no-cpu-baseline --no-extra > $O/bench_m$m.json 2>/dev/null
done
head -14 $O/kernel_stats_m1.csv; head -14 $O/kernel_stats_m0.csv
