# A/B of library build variants: bash tools/scratch/ab_libs.sh "<kernel name pattern>" lib1.so lib2.so ...
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
pat=$1; shift
for lib in "$@"; do
  for rep in 1 2; do
  echo -n "$lib "; PQA_LIB=$PWD/$lib python bench.py --no-cpu-baseline --no-extra --no-profile 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],2))"
  done
  d=/tmp/ab_$(basename $lib .so); rm -rf $d
  PQA_LIB=$PWD/$lib rocprofv3 --kernel-trace --stats -d $d -o r -- python bench.py --no-cpu-baseline --no-extra --no-profile --steps 4 > /dev/null 2>&1
  python tools/prof_stats.py $d/r_results.db 2>/dev/null | grep -E "$pat" | cut -c1-60,100-160
done
