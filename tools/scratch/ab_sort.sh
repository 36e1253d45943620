cd $GRAFT_REPO_ROOT
PQA_ORB_SORT_MIN=1 PQA_ORB_WIDE=0 timeout 600 python -m pytest tests/test_gpu_pbc.py -x -q -m gpu 2>&1 | tail -2
for sm in 1000000000 65536 16384; do
  for cfg in "c5 4096" "c5 16384" "c3 32768"; do set -- $cfg
    echo -n "sort_min=$sm $1@$2 "; PQA_ORB_SORT_MIN=$sm timeout 120 python tools/config_bench.py $1 --walkers $2 --steps 8 2>/dev/null | tail -1 | sed 's/.*ms_per_step/ms_per_step/' | cut -c1-75
  done
  echo -n "sort_min=$sm k222@32768 "; PQA_ORB_SORT_MIN=$sm timeout 120 python tools/pbc_bench.py --case k222 --walkers 32768 --steps 3 2>/dev/null | tail -1 | sed 's/.*ms_per_step/ms_per_step/' | cut -c1-75
done
