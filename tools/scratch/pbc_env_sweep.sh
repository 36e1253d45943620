R=$GRAFT_REPO_ROOT; cd $R
one() { python tools/config_bench.py $1 --walkers $2 --steps $3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), end='  ')"; }
for v in "PQA_NONE=1" "PQA_ORB_KC5=16" "PQA_ORB_TP=64" "PQA_ORB_KC5=16 PQA_ORB_TP=64"; do
  echo -n "$v : c5@4096 "; env $v bash -c "$(declare -f one); one c5 4096 10"; echo -n " c5@16384 "; env $v bash -c "$(declare -f one); one c5 16384 6"; echo -n " c5@32768 "; env $v bash -c "$(declare -f one); one c5 32768 4"
  echo -n " c3@8192 "; env $v bash -c "$(declare -f one); one c3 8192 8"; echo -n " c3@32768 "; env $v bash -c "$(declare -f one); one c3 32768 4"; echo
done
