# kernel stats of the (H2O)8 step at a walker count: bash tools/scratch/prof_w.sh 16384
R=$GRAFT_REPO_ROOT; W=${1:-16384}; cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/pw; rocprofv3 --kernel-trace --stats -d /tmp/pw -o d -- python $R/tools/scratch/lib_bench.py $R/pyqmc_amd/lib/libpyqmc_amd.so $W > /dev/null 2>&1 < /dev/null
python $R/tools/prof_stats.py /tmp/pw/d_results.db /tmp/pw/stats.csv; head -9 /tmp/pw/stats.csv | cut -c1-140
