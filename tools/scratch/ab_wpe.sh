cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for w in 4 6 8; do for gm in 4 8; do
  lib=pyqmc_amd/lib/ab/libpqa_wpe$w.so
  echo -n "wpe=$w gm=$gm "; PQA_LW_GM=$gm PQA_LIB=$PWD/$lib python bench.py --no-cpu-baseline --no-extra --no-profile 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],2))"
  d=/tmp/ab_$w_$gm; rm -rf $d
  PQA_LW_GM=$gm PQA_LIB=$PWD/$lib rocprofv3 --kernel-trace --stats -d $d -o r -- python bench.py --no-cpu-baseline --no-extra --no-profile --steps 4 > /dev/null 2>&1
  python tools/prof_stats.py $d/r_results.db 2>/dev/null | grep -E "move_part|fin_lw" | sed 's/(SysDev[^"]*"/"/' | cut -c1-90
done; done
