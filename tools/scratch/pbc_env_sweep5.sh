R=$GRAFT_REPO_ROOT; cd $R
for v in "PQA_NONE=1" "PQA_PRE_GRID=8" "PQA_PRE_GRID=12" "PQA_PRE_GRID=16" "PQA_WIDE_NTH=512" "PQA_STEP_GW=16" "PQA_TM_PRE=0" "PQA_ECP_ATOM_MAJOR=0" "PQA_NONE=1"; do
  echo -n "$v : "
  for w in 4096 16384; do env $v python tools/config_bench.py c5 --walkers $w --steps 10 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c5@$w', round(d['ms_per_step'],2), end='  ')"; done
  env $v python tools/pbc_bench.py --case k222 --walkers 8192 --steps 4 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('k222@8192', round(d['ms_per_step'],2), end='  ')"; echo
done
