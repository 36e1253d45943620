# kernel stats + bench line of the headline at HEAD: bash tools/scratch/quick_prof.sh [tag] [extra pytest -k expression]
R=$GRAFT_REPO_ROOT; T=${1:-q}; O=$R/gpurun_out/quick_$T; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/pbq; rocprofv3 --kernel-trace --stats -d /tmp/pbq -o b -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra > $O/bench_prof.json 2>/dev/null < /dev/null
python $R/tools/prof_stats.py /tmp/pbq/b_results.db $O/kernel_stats.csv
python $R/bench.py --no-cpu-baseline --no-extra > $O/bench.json 2>/dev/null
head -9 $O/kernel_stats.csv | cut -c1-150; python -c "import json;d=json.load(open('$O/bench.json'));print(d['value'],d['ms_per_step'])"
if [ -n "$2" ]; then (cd $R && python -m pytest tests -m gpu -x -q -k "$2" 2>&1 | tail -4); fi
