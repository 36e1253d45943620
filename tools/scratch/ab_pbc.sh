# periodic orbital path: tests + the periodic bench lines
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -m pytest tests/test_gpu_pbc.py -x -q -m gpu 2>&1 | tail -3
for w in 8192 32768; do python tools/pbc_bench.py --case k222 --walkers $w --steps 3 2>/dev/null | tail -1 | cut -c1-140; done
for w in 8192; do python tools/pbc_bench.py --case cubic --walkers $w --steps 3 2>/dev/null | tail -1 | cut -c1-140; done
python tools/config_bench.py c5 --walkers 4096 --steps 10 2>/dev/null | tail -1 | cut -c1-160
python tools/config_bench.py c5 --walkers 16384 --steps 6 2>/dev/null | tail -1 | cut -c1-160
python tools/config_bench.py c3 --walkers 8192 --steps 4 2>/dev/null | tail -1 | cut -c1-160
rm -rf /tmp/pk; rocprofv3 --kernel-trace --stats -d /tmp/pk -o k -- python tools/pbc_bench.py --case k222 --walkers 32768 --steps 3 > /dev/null 2>&1 < /dev/null
python tools/prof_stats.py /tmp/pk/k_results.db | head -8 | sed 's/(SysDev[^"]*"/"/' | cut -c1-100
