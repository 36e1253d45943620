R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --pmc $set -d /tmp/lp_$i -o t -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile > /dev/null 2>&1 < /dev/null
  python $R/tools/pmc_counters.py /tmp/lp_$i/t_results.db 2>&1 | grep "k_move_part_lw\|k_commit_lw\|k_accept_fin"
done
