cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for v in pyqmc_amd/lib/libpyqmc_amd.so pyqmc_amd/lib/variants/expneg.so; do
  echo "== $v"
  for w in 4096 65536; do echo -n "M W=$w "; python tools/scratch/lib_bench.py $v $w; done
  rm -rf /tmp/pp; rocprofv3 --kernel-trace --stats -d /tmp/pp -o k -- python tools/scratch/lib_bench.py $v 65536 > /tmp/pp.log 2>&1 < /dev/null
  python tools/prof_stats.py /tmp/pp/k_results.db | grep "k_orb<" | cut -c1-130
  rm -rf /tmp/pp; rocprofv3 --kernel-trace --stats -d /tmp/pp -o k -- python tools/scratch/r3_pre_abl.py $v 32768 > /tmp/pp.log 2>&1 < /dev/null
  tail -1 /tmp/pp.log; python tools/prof_stats.py /tmp/pp/k_results.db | grep "k_orb_wide\|k_orb<" | cut -c1-130
done
