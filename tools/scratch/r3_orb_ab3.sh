cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r3_orb; mkdir -p $O; rm -f $O/ab3.txt
for rep in 1 2; do for v in lib_unif.so lib_unif_u2.so lib_unif_u3.so; do for nt in 0 1; do
  echo -n "NOTAB=$nt " >> $O/ab3.txt; PQA_ORB_NOTAB=$nt timeout 300 python tools/orb_time.py variants/$v 65536 2>/dev/null | tail -1 >> $O/ab3.txt
done; done; done
for v in lib_unif.so lib_unif_u2.so lib_unif_u3.so; do for nt in 0 1; do
  echo -n "NOTAB=$nt $v step " >> $O/ab3.txt; PQA_ORB_NOTAB=$nt timeout 300 python tools/scratch/lib_bench.py pyqmc_amd/lib/variants/$v 65536 >> $O/ab3.txt 2>&1
done; done
cat $O/ab3.txt
