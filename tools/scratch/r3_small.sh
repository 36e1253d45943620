# small-shard A/B: env toggles on the head library + the 16-walker tile variant
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r3_small; mkdir -p $O; rm -f $O/*.txt
L=pyqmc_amd/lib/libpyqmc_amd.so
for w in 2048 4096 8192 16384; do
  for cfg in "PQA_STEP_PRE=1 PQA_DRAWS_MAX=16384" "PQA_STEP_PRE=0 PQA_DRAWS_MAX=16384" "PQA_STEP_PRE=1 PQA_DRAWS_MAX=0" "PQA_STEP_PRE=0 PQA_DRAWS_MAX=0"; do
    echo -n "W=$w $cfg " >> $O/ab.txt; env $cfg timeout 300 python tools/scratch/lib_bench.py $L $w >> $O/ab.txt 2>&1
  done
  echo -n "W=$w tile12 " >> $O/ab.txt; PQA_LW=2 timeout 300 python tools/scratch/lib_bench.py $L $w >> $O/ab.txt 2>&1
  echo -n "W=$w tile16 " >> $O/ab.txt; PQA_LW=2 timeout 300 python tools/scratch/lib_bench.py pyqmc_amd/lib/variants/tile16.so $w >> $O/ab.txt 2>&1
done
cat $O/ab.txt
