cd $GRAFT_REPO_ROOT
for cfg in "c5 1024" "c5 4096" "c5 8192"; do set -- $cfg
  for nth in 512 1024; do echo -n "$1@$2 nth=$nth "; PQA_WIDE_NTH=$nth timeout 120 python tools/config_bench.py $1 --walkers $2 --steps 10 2>/dev/null | tail -1 | sed 's/.*ms_per_step/ms_per_step/' | cut -c1-80; done
done
for w in 4096 8192; do echo -n "c3@$w "; timeout 120 python tools/config_bench.py c3 --walkers $w --steps 8 2>/dev/null | tail -1 | sed 's/.*ms_per_step/ms_per_step/' | cut -c1-80; done
timeout 300 python -m pytest tests/test_gpu_pbc.py -x -q -m gpu 2>&1 | tail -2
PQA_WIDE_NTH=1024 timeout 300 python -m pytest tests/test_gpu_pbc.py -x -q -m gpu 2>&1 | tail -2
