# where does the periodic orbital kernel spend its time?  compile-time ablations (pyqmc_amd/lib/ab/libpqa_<X>.so):
#   NOP1 no AO phase at all; NOWALKZERO per-shell set-up only (context, tables); NOWALK + zeroing of the shell's tile rows;
#   NOADD + image walk without evaluating any shell; NOEXP shells evaluated with exp replaced by a linear term
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for lib in pyqmc_amd/lib/libpyqmc_amd.so "$@"; do
  rm -rf /tmp/pk; PQA_ORB_TP=32 PQA_LIB=$PWD/$lib timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/pk -o k -- python tools/pbc_bench.py --case k222 --walkers 32768 --steps 2 > /dev/null 2>&1 < /dev/null
  echo "== $lib"; python tools/prof_stats.py /tmp/pk/k_results.db | grep -E "k_orb<" | sed 's/(SysDev[^"]*"/"/' | cut -c1-90
done
