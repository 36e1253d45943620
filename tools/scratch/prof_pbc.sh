cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf /tmp/pc; timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/pc -o k -- python tools/pbc_bench.py --case $1 --walkers $2 --steps 3 > /tmp/pc.log 2>&1 < /dev/null
grep ms_per_step /tmp/pc.log | sed 's/.*ms_per_step/ms_per_step/' | cut -c1-60
python tools/prof_stats.py /tmp/pc/k_results.db | head -${3:-8} | sed 's/(SysDev[^"]*"/"/' | cut -c1-100
