cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for cfg in "" "PQA_ORB_WIDE_MAX=1 PQA_ORB_TP=16" "PQA_ORB_WIDE_MAX=1 PQA_ORB_TP=32" "PQA_ORB_WIDE_MAX=1 PQA_ORB_TP=64"; do
  rm -rf /tmp/pk; env $cfg timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/pk -o k -- python tools/pbc_bench.py --case k222 --walkers 32768 --steps 2 > /tmp/pb.out 2>/dev/null < /dev/null
  echo "== $cfg"; python tools/prof_stats.py /tmp/pk/k_results.db | grep -E "k_orb<5|k_orb_wide|k_pbc_prepass" | sed 's/(SysDev[^"]*"/"/' | cut -c1-90 | head -4
  tail -1 /tmp/pb.out | cut -c1-150
done
