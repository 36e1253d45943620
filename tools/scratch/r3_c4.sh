cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r3_c4; mkdir -p $O; rm -f $O/*
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
for w in 2048 16384; do python tools/config_bench.py c4 --walkers $w --steps 4 2>/dev/null | tail -1 | cut -c1-200; done
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/pd; rocprofv3 --kernel-trace --stats -d /tmp/pd -o d -- python $GRAFT_REPO_ROOT/tools/config_bench.py c4 --walkers 2048 --steps 4 > /dev/null 2>&1 < /dev/null
python $GRAFT_REPO_ROOT/tools/prof_stats.py /tmp/pd/d_results.db $O/c4_2048_kernel_stats.csv; head -9 $O/c4_2048_kernel_stats.csv | cut -c1-150
