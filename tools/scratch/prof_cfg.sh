# kernel summary of one configuration: bash tools/scratch/prof_cfg.sh c3 8192 4
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf /tmp/pc; rocprofv3 --kernel-trace --stats -d /tmp/pc -o k -- python tools/config_bench.py $1 --walkers $2 --steps $3 > /tmp/pc.log 2>&1 < /dev/null
tail -1 /tmp/pc.log | cut -c1-200
python tools/prof_stats.py /tmp/pc/k_results.db | head -${4:-10} | sed 's/(SysDev[^"]*"/"/' | cut -c1-100
