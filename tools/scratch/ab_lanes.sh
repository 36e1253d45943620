cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_pbc.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -2
for cfg in "c5 1024" "c5 4096" "c5 8192" "c3 4096" "c3 8192"; do set -- $cfg
  for lm in 0 8192 16384; do echo -n "$1@$2 lanes_max=$lm "; PQA_PRE_LANES_MAX=$lm timeout 120 python tools/config_bench.py $1 --walkers $2 --steps 10 2>/dev/null | tail -1 | sed 's/.*ms_per_step/ms_per_step/' | cut -c1-80; done
done
