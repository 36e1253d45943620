# round 3: the other BASELINE configurations + periodic VMC at HEAD, with kernel stats for C5 and C4
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r3_${1:-configs}; mkdir -p $O; rm -f $O/*.jsonl
python tools/config_bench.py c2 --walkers 4096 --steps 20 2>/dev/null | tail -1 >> $O/config_bench.jsonl
python tools/config_bench.py c3 --walkers 8192 --steps 8 2>/dev/null | tail -1 >> $O/config_bench.jsonl
python tools/config_bench.py c4 --walkers 2048 --steps 4 2>/dev/null | tail -1 >> $O/config_bench.jsonl
python tools/config_bench.py c4 --walkers 16384 --steps 4 2>/dev/null | tail -1 >> $O/config_bench.jsonl
python tools/config_bench.py c5 --walkers 4096 --steps 10 2>/dev/null | tail -1 >> $O/config_bench.jsonl
python tools/config_bench.py c5 --walkers 16384 --steps 10 2>/dev/null | tail -1 >> $O/config_bench.jsonl
for c in k222 cubic; do for w in 8192 32768; do python tools/pbc_bench.py --case $c --walkers $w --steps 4 2>/dev/null | tail -1 >> $O/pbc_bench.jsonl; done; done
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/pd; rocprofv3 --kernel-trace --stats -d /tmp/pd -o d -- python $GRAFT_REPO_ROOT/tools/config_bench.py c5 --walkers 4096 --steps 10 > /dev/null 2>&1 < /dev/null
python $GRAFT_REPO_ROOT/tools/prof_stats.py /tmp/pd/d_results.db $O/dmc_c5_4096_kernel_stats.csv
rm -rf /tmp/pd; rocprofv3 --kernel-trace --stats -d /tmp/pd -o d -- python $GRAFT_REPO_ROOT/tools/config_bench.py c4 --walkers 2048 --steps 4 > /dev/null 2>&1 < /dev/null
python $GRAFT_REPO_ROOT/tools/prof_stats.py /tmp/pd/d_results.db $O/c4_2048_kernel_stats.csv
cat $O/config_bench.jsonl $O/pbc_bench.jsonl | cut -c1-420; head -16 $O/dmc_c5_4096_kernel_stats.csv; head -12 $O/c4_2048_kernel_stats.csv
