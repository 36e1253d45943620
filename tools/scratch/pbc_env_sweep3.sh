R=$GRAFT_REPO_ROOT; cd $R
for v in "PQA_NONE=1" "PQA_ORB_WIDE_MAX=16384" "PQA_ORB_WIDE_MAX=32768" "PQA_ORB_WIDE_MAX=70000" "PQA_ORB_SPLIT_MAX=0" "PQA_ORB_SPLIT_MAX=32768" "PQA_LW_KB=8" "PQA_LW_KB=4"; do
  echo -n "$v : "
  for w in 4096 16384; do env $v python tools/config_bench.py c5 --walkers $w --steps 10 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c5@$w', round(d['ms_per_step'],2), end='  ')"; done
  for w in 8192 32768; do env $v python tools/pbc_bench.py --case k222 --walkers $w --steps 4 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('k222@$w', round(d['ms_per_step'],2), end='  ')"; done
  env $v python tools/config_bench.py c3 --walkers 8192 --steps 8 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c3@8192', round(d['ms_per_step'],2), end='  ')"; echo
done
