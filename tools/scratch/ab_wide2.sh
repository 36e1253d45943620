cd $GRAFT_REPO_ROOT
for w in 1024 8192 16384 32768; do for wide in 1 0; do
  echo -n "W=$w PQA_ORB_WIDE=$wide "; PQA_ORB_WIDE=$wide python bench.py --walkers $w --no-cpu-baseline --no-extra --no-profile 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],2))"
done; done
for wide in 1 0; do echo -n "c5@1024 wide=$wide "; PQA_ORB_WIDE=$wide python tools/config_bench.py c5 --walkers 1024 --steps 10 2>/dev/null | tail -1 | cut -c100-170; done
for wide in 1 0; do echo -n "c5@8192 wide=$wide "; PQA_ORB_WIDE=$wide python tools/config_bench.py c5 --walkers 8192 --steps 6 2>/dev/null | tail -1 | cut -c100-170; done
