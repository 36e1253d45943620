cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
python tools/config_bench.py c5 --walkers 4096 --steps 10 2>/dev/null | tail -1 | cut -c1-250
python tools/config_bench.py c3 --walkers 8192 --steps 8 2>/dev/null | tail -1 | cut -c1-250
rm -rf /tmp/pd; rocprofv3 --kernel-trace --stats -d /tmp/pd -o d -- python tools/config_bench.py c5 --walkers 4096 --steps 10 > /dev/null 2>&1 < /dev/null
python tools/prof_stats.py /tmp/pd/d_results.db gpurun_out/c5_4096_now.csv; head -24 gpurun_out/c5_4096_now.csv | sed 's/(SysDev[^"]*"/"/' | cut -c1-110
