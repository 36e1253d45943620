# Regenerates the measured evidence under gpurun_out/evidence (copy what is to be judged into profiles/):
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash tools/refresh_evidence.sh'
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/evidence; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
python $R/bench.py > $O/bench.json 2> $O/bench.err < /dev/null
rocprofv3 --kernel-trace --stats -d /tmp/pb -o b -- python $R/bench.py --steps 10 --warmup 2 > /dev/null 2>&1 < /dev/null
python $R/tools_prof.py /tmp/pb/b_results.db $O/bench_kernel_stats.csv
for c in k222 cubic; do for w in 8192 32768; do python $R/tools/pbc_bench.py --case $c --walkers $w --steps 4 2>/dev/null | tail -1 >> $O/pbc_bench.jsonl; done; done
rocprofv3 --kernel-trace --stats -d /tmp/pk -o k -- python $R/tools/pbc_bench.py --case k222 --walkers 32768 --steps 3 > /dev/null 2>&1 < /dev/null
python $R/tools_prof.py /tmp/pk/k_results.db $O/pbc_k222_kernel_stats.csv
python $R/tools/config_bench.py c2 --walkers 4096 --steps 20 2>/dev/null | tail -1 >> $O/config_bench.jsonl
python $R/tools/config_bench.py c2 --walkers 65536 --steps 20 2>/dev/null | tail -1 >> $O/config_bench.jsonl
python $R/tools/config_bench.py c3 --walkers 8192 --steps 4 2>/dev/null | tail -1 >> $O/config_bench.jsonl
python $R/tools/config_bench.py c3 --walkers 32768 --steps 4 2>/dev/null | tail -1 >> $O/config_bench.jsonl
python $R/tools/config_bench.py c4 --walkers 2048 --steps 4 2>/dev/null | tail -1 >> $O/config_bench.jsonl
python $R/tools/config_bench.py c4 --walkers 16384 --steps 4 2>/dev/null | tail -1 >> $O/config_bench.jsonl
python $R/tools/config_bench.py c5 --walkers 4096 --steps 10 2>/dev/null | tail -1 >> $O/config_bench.jsonl
python $R/tools/config_bench.py c5 --walkers 16384 --steps 10 2>/dev/null | tail -1 >> $O/config_bench.jsonl
python $R/tools/config_bench.py c5 --walkers 32768 --steps 10 2>/dev/null | tail -1 >> $O/config_bench.jsonl
python $R/tools/config_bench.py c5 --walkers 256 --steps 1 --host 2>/dev/null | tail -1 >> $O/config_bench.jsonl
rocprofv3 --kernel-trace --stats -d /tmp/pd -o d -- python $R/tools/config_bench.py c5 --walkers 16384 --steps 10 > /dev/null 2>&1 < /dev/null
python $R/tools_prof.py /tmp/pd/d_results.db $O/dmc_c5_kernel_stats.csv
ls -la $O
