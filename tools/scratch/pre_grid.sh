# near-candidate masks of the periodic pre-pass (PQA_PRE_GRID): kernel time of k_pbc_prepass with and without, per launch size
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
{
for cs in "k222 32768" "k222 4096" "cubic 8192"; do
  set -- $cs
  for g in 0 8 16; do
    rm -rf /tmp/pk; PQA_PRE_GRID=$g timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/pk -o k -- python tools/pbc_bench.py --case $1 --walkers $2 --steps 2 > /tmp/pb.out 2>/dev/null < /dev/null
    echo "== $1 $2 walkers, PQA_PRE_GRID=$g"; python tools/prof_stats.py /tmp/pk/k_results.db | grep -E "k_orb<5|k_orb_wide|k_pbc_prepass" | sed 's/(SysDev[^"]*"/"/' | cut -c1-90
    tail -1 /tmp/pb.out | cut -c1-200
  done
done
} > gpurun_out/r4_pre_grid.txt 2>&1
cat gpurun_out/r4_pre_grid.txt
