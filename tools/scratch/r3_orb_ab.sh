# A/B of orbital-kernel variants: usage r3_orb_ab.sh libA libB ...
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r3_orb; mkdir -p $O
for rep in 1 2; do for v in "$@"; do for n in 65536 32768; do
  timeout 300 python tools/orb_time.py variants/$v $n 2>/dev/null | tail -1 | sed "s/^/n=$n /" >> $O/ab.txt
done; done; done
for v in "$@"; do
  echo -n "$v step " >> $O/ab.txt; timeout 300 python tools/scratch/lib_bench.py pyqmc_amd/lib/variants/$v 65536 >> $O/ab.txt 2>&1
  echo -n "$v step4096 " >> $O/ab.txt; timeout 300 python tools/scratch/lib_bench.py pyqmc_amd/lib/variants/$v 4096 >> $O/ab.txt 2>&1
done
cat $O/ab.txt
