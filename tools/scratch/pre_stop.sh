cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for st in 1 2 3 0; do
rm -rf /tmp/pc; PQA_PRE_STOP=$st timeout 120 rocprofv3 --kernel-trace --stats -d /tmp/pc -o k -- python tools/config_bench.py c5 --walkers ${1:-4096} --steps 3 > /tmp/pc.log 2>&1 < /dev/null
echo -n "stop=$st "; python tools/prof_stats.py /tmp/pc/k_results.db | grep prepass | sed "s/(SysDev[^\"]*\"/\"/"
done
