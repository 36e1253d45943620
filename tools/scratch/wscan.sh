# walkers-per-GPU scan of the two lane-per-walker sweeps: bash tools/scratch/wscan.sh > gpurun_out/wscan.txt
cd $GRAFT_REPO_ROOT
for lw in 3 1; do for w in 1024 4096 16384 65536; do
  echo -n "PQA_LW=$lw W=$w "; PQA_LW=$lw python bench.py --walkers $w --no-cpu-baseline --steps 8 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), d['ms_per_step'])"
done; done
