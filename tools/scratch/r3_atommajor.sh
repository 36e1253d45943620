cd $GRAFT_REPO_ROOT; L=pyqmc_amd/lib/libpyqmc_amd.so
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -3
for a in 1 0; do
  echo "PQA_ECP_ATOM_MAJOR=$a"
  for w in 4096 65536; do echo -n "M W=$w "; PQA_ECP_ATOM_MAJOR=$a python tools/scratch/lib_bench.py $L $w; done
  PQA_ECP_ATOM_MAJOR=$a python tools/config_bench.py c5 --walkers 4096 --steps 10 2>/dev/null | tail -1 | cut -c60-170
  PQA_ECP_ATOM_MAJOR=$a python tools/config_bench.py c3 --walkers 8192 --steps 8 2>/dev/null | tail -1 | cut -c60-170
  PQA_ECP_ATOM_MAJOR=$a python tools/pbc_bench.py --case k222 --walkers 32768 --steps 4 2>/dev/null | tail -1 | cut -c1-130
done
