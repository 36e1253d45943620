# round 3: where k_step_lw's time goes (compile-time ablations) + block geometry scan
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r3_abl; mkdir -p $O
timeout 600 python tools/scratch/r3_fuse_check.py 4096 > $O/check.txt 2>&1
timeout 600 python tools/scratch/r3_fuse_check.py 20000 >> $O/check.txt 2>&1
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt >> $O/check.txt
for w in 1024 2048 4096 8192 16384 32768 65536; do
  echo -n "default W=$w " >> $O/geom.txt; timeout 300 python tools/scratch/lib_bench.py pyqmc_amd/lib/libpyqmc_amd.so $w >> $O/geom.txt 2>&1
done
for w in 4096 16384 65536; do for gm in 2 4 8 16; do
  echo -n "GM=$gm W=$w " >> $O/geom.txt; PQA_LW_GM=$gm timeout 300 python tools/scratch/lib_bench.py pyqmc_amd/lib/libpyqmc_amd.so $w >> $O/geom.txt 2>&1
done; done
cd /tmp; export TMPDIR=/tmp
for v in base norefresh nocommit nojas noslater; do for w in 65536 4096; do
  rm -rf /tmp/pp; rocprofv3 --kernel-trace --stats -d /tmp/pp -o b -- python $GRAFT_REPO_ROOT/tools/scratch/lib_bench.py $GRAFT_REPO_ROOT/pyqmc_amd/lib/variants/lib_$v.so $w > /tmp/pp.log 2>&1 < /dev/null
  python $GRAFT_REPO_ROOT/tools/prof_stats.py /tmp/pp/b_results.db /tmp/pp/s.csv
  echo "== $v W=$w $(tail -1 /tmp/pp.log)" >> $O/abl.txt; grep -E "k_step_lw|k_orb|k_flush" /tmp/pp/s.csv | cut -c1-60,100- >> $O/abl.txt
done; done
cat $O/check.txt $O/geom.txt $O/abl.txt
