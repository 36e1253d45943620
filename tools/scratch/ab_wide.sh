# A/B of the whole-K small-launch orbital kernel: bash tools/scratch/ab_wide.sh
cd $GRAFT_REPO_ROOT
for wide in -1 0; do
  echo "== PQA_ORB_WIDE=$wide"
  export PQA_ORB_WIDE=$wide
  python tools/config_bench.py c5 --walkers 4096 --steps 10 2>/dev/null | tail -1 | cut -c60-170
  python tools/config_bench.py c3 --walkers 8192 --steps 4 2>/dev/null | tail -1 | cut -c60-170
  for c in k222 cubic; do python tools/pbc_bench.py --case $c --walkers 8192 --steps 4 2>/dev/null | tail -1 | cut -c1-130; done
  python tools/config_bench.py c2 --walkers 4096 --steps 20 2>/dev/null | tail -1 | cut -c60-170
  python bench.py --walkers 4096 --no-cpu-baseline --no-extra --no-profile 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('M@4096', round(d['value']), round(d['ms_per_step'],2))"
done
