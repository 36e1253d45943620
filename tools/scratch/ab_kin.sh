# kinetic kernel: electrons per block (PQA_KIN_EB build variants in pyqmc_amd/lib/ab/), wave-uniform electron index
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for lib in pyqmc_amd/lib/libpyqmc_amd.so pyqmc_amd/lib/ab/libpqa_eb2.so pyqmc_amd/lib/ab/libpqa_eb4.so pyqmc_amd/lib/ab/libpqa_eb8.so; do
  for rep in 1 2; do
  echo -n "$lib "; PQA_LIB=$PWD/$lib python bench.py --no-cpu-baseline --no-extra --no-profile 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],2))"
  done
  d=/tmp/abkin_$(basename $lib .so); rm -rf $d
  PQA_LIB=$PWD/$lib rocprofv3 --kernel-trace --stats -d $d -o r -- python bench.py --no-cpu-baseline --no-extra --no-profile --steps 4 > /dev/null 2>&1
  python tools/prof_stats.py $d/r_results.db 2>/dev/null | grep -i "kinetic_lw" | cut -c1-140
done
