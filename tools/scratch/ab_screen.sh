# A/B of primitive screening: bash tools/scratch/ab_screen.sh  (needs pyqmc_amd/lib/libpyqmc_amd_noscreen.so)
cd $GRAFT_REPO_ROOT
for lib in "" pyqmc_amd/lib/libpyqmc_amd_noscreen.so; do
  echo "== PQA_LIB=$lib"
  PQA_LIB=$lib python bench.py --no-cpu-baseline --no-extra 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('M', round(d['value']), round(d['ms_per_step'],2), 'orb us', round(1e3*d['roofline']['avg_launch_ms'],1))"
  for c in k222 cubic; do PQA_LIB=$lib python tools/pbc_bench.py --case $c --walkers 8192 --steps 4 2>/dev/null | tail -1 | cut -c1-130; done
  PQA_LIB=$lib python tools/config_bench.py c5 --walkers 4096 --steps 10 2>/dev/null | tail -1 | cut -c60-170
  PQA_LIB=$lib python tools/config_bench.py c3 --walkers 8192 --steps 4 2>/dev/null | tail -1 | cut -c60-170
done
