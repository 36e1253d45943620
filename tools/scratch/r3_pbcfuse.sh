cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r3_pbcfuse; mkdir -p $O; rm -f $O/*
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
for f in 1 0; do
  echo "PQA_PBC_FUSE=$f"
  PQA_PBC_FUSE=$f python tools/config_bench.py c5 --walkers 4096 --steps 10 2>/dev/null | tail -1 | cut -c1-170
  PQA_PBC_FUSE=$f python tools/config_bench.py c3 --walkers 8192 --steps 8 2>/dev/null | tail -1 | cut -c1-170
  for c in k222 cubic; do for w in 8192 32768; do PQA_PBC_FUSE=$f python tools/pbc_bench.py --case $c --walkers $w --steps 4 2>/dev/null | tail -1 | cut -c1-130; done; done
done
