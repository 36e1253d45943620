R=$GRAFT_REPO_ROOT; cd $R
for v in "PQA_NONE=1" "PQA_ORB_KC1=16" "PQA_NONE=1" "PQA_ORB_KC1=16"; do
  echo -n "$v : "
  for c in k222 cubic; do for w in 8192 32768; do env $v python tools/pbc_bench.py --case $c --walkers $w --steps 4 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$c@$w', round(d['ms_per_step'],2), end='  ')"; done; done
  for w in 4096 8192 32768; do env $v python tools/config_bench.py c3 --walkers $w --steps 8 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c3@$w', round(d['ms_per_step'],2), end='  ')"; done; echo
done
