cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for lib in pyqmc_amd/lib/ab/libpqa_nop1.so; do
rm -rf /tmp/pk; PQA_LIB=$PWD/$lib rocprofv3 --kernel-trace --stats -d /tmp/pk -o k -- python tools/pbc_bench.py --case k222 --walkers 32768 --steps 2 > /dev/null 2>&1 < /dev/null
python tools/prof_stats.py /tmp/pk/k_results.db | head -6 | sed 's/(SysDev[^"]*"/"/' | cut -c1-100
done
