cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r3_kb; mkdir -p $O; rm -f $O/*
L=pyqmc_amd/lib/libpyqmc_amd.so
for rep in 1 2; do for w in 65536 16384; do for kb in 4 5 6 8; do
  echo -n "W=$w PQA_LW_KB=$kb " >> $O/ab.txt; PQA_LW_KB=$kb timeout 300 python tools/scratch/lib_bench.py $L $w >> $O/ab.txt 2>&1
done; done; done
sort $O/ab.txt
