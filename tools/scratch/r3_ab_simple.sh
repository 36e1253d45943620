# step-time A/B of library variants at several walker counts: r3_ab_simple.sh "W1 W2 ..." lib1.so lib2.so ...
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r3_ab; mkdir -p $O; rm -f $O/ab.txt; WS="$1"; shift
for rep in 1 2; do for v in "$@"; do for w in $WS; do
  echo -n "W=$w " >> $O/ab.txt; timeout 300 python tools/scratch/lib_bench.py pyqmc_amd/lib/variants/$v $w >> $O/ab.txt 2>&1
done; done; done
sort $O/ab.txt
