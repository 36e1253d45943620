# wide orbital kernel for periodic cells after the image-list change: forced on (1) / off (0) / automatic (unset)
cd $GRAFT_REPO_ROOT
for cfg in "c3 4096" "c3 8192" "c5 1024" "c5 4096" "c5 8192"; do set -- $cfg
  for wide in 1 0; do echo -n "$1@$2 wide=$wide "; PQA_ORB_WIDE=$wide timeout 120 python tools/config_bench.py $1 --walkers $2 --steps 6 2>/dev/null | tail -1 | sed 's/.*ms_per_step/ms_per_step/' | cut -c1-80; done
done
for w in 4096 8192; do for wide in 1 0; do echo -n "cubic@$w wide=$wide "; PQA_ORB_WIDE=$wide timeout 120 python tools/pbc_bench.py --case cubic --walkers $w --steps 4 2>/dev/null | tail -1 | sed 's/.*ms_per_step/ms_per_step/' | cut -c1-80; done; done
