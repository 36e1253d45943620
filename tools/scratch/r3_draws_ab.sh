cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r3_draws; mkdir -p $O; rm -f $O/*
L=pyqmc_amd/lib/libpyqmc_amd.so
for rep in 1 2; do for w in 16384 32768 65536; do for d in 16384 1000000; do
  echo -n "W=$w PQA_DRAWS_MAX=$d " >> $O/ab.txt; PQA_DRAWS_MAX=$d timeout 300 python tools/scratch/lib_bench.py $L $w >> $O/ab.txt 2>&1
done; done; done
sort $O/ab.txt
