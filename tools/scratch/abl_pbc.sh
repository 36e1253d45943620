# where does the periodic orbital kernel spend its time?  compile-time ablations, built with
#   python -c "import __graft_entry__ as g, os; [g.build(extra_flags=['-DPQA_ABL_' + v], lib=os.path.join(g.LIBDIR, 'libpqa_' + v + '.so')) for v in ('NOEXP', 'NOADD', 'NOWALK', 'NOP1')]"
#   NOP1 no AO phase at all; NOWALK+NOZERO per-shell set-up only (context, tables); NOWALK + zeroing of the shell's tile rows;
#   NOADD + image walk without evaluating any shell; NOEXP shells evaluated with exp replaced by a linear term
#   bash tools/scratch/abl_pbc.sh pyqmc_amd/lib/libpqa_NOADD.so ...      (the launch's own tile choice: k_orb_wide at 32768 points)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for lib in pyqmc_amd/lib/libpyqmc_amd.so "$@"; do
  rm -rf /tmp/pk; PQA_LIB=$PWD/$lib timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/pk -o k -- python tools/pbc_bench.py --case k222 --walkers 32768 --steps 2 > /dev/null 2>&1 < /dev/null
  echo "== $lib"; python tools/prof_stats.py /tmp/pk/k_results.db | grep -E "k_orb<5|k_orb_wide|k_pbc_prepass" | sed 's/(SysDev[^"]*"/"/' | cut -c1-90
done
