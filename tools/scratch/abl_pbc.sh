# where does the periodic orbital kernel spend its time?  compile-time ablations, built with
#   for v in NOEXP NOADD NOWALK "NOWALK -DPQA_ABL_NOZERO" NOP1; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DPQA_ABL_$v pyqmc_amd/csrc/pqa_capi.hip -o pyqmc_amd/lib/ab/libpqa_<name>.so; done
#   NOP1 no AO phase at all; NOWALK+NOZERO per-shell set-up only (context, tables); NOWALK + zeroing of the shell's tile rows;
#   NOADD + image walk without evaluating any shell; NOEXP shells evaluated with exp replaced by a linear term
#   bash tools/scratch/abl_pbc.sh pyqmc_amd/lib/ab/libpqa_NOADD.so ...
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for lib in pyqmc_amd/lib/libpyqmc_amd.so "$@"; do
  rm -rf /tmp/pk; PQA_ORB_TP=32 PQA_LIB=$PWD/$lib timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/pk -o k -- python tools/pbc_bench.py --case k222 --walkers 32768 --steps 2 > /dev/null 2>&1 < /dev/null
  echo "== $lib"; python tools/prof_stats.py /tmp/pk/k_results.db | grep -E "k_orb<" | sed 's/(SysDev[^"]*"/"/' | cut -c1-90
done
