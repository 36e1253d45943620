cd /tmp; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r3_pbcprof; mkdir -p $O
for w in 32768 8192; do
rm -rf /tmp/pk; rocprofv3 --kernel-trace --stats -d /tmp/pk -o k -- python $GRAFT_REPO_ROOT/tools/pbc_bench.py --case k222 --walkers $w --steps 3 > /tmp/pk.log 2>&1 < /dev/null
python $GRAFT_REPO_ROOT/tools/prof_stats.py /tmp/pk/k_results.db $O/k222_${w}_kernel_stats.csv
tail -1 /tmp/pk.log | cut -c1-200; head -14 $O/k222_${w}_kernel_stats.csv | cut -c1-70,150-
done
cd $GRAFT_REPO_ROOT; timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
