R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_MFMA"; do
  i=$((i+1))
  rocprofv3 --pmc $set -d /tmp/tp_$i -o t -- python $R/tools/scratch/tile_time.py $R/tools/scratch/lib_BASE.so 16384 > /dev/null 2>&1 < /dev/null
  python $R/tools/pmc_counters.py /tmp/tp_$i/t_results.db 2>&1 | grep sweep_tile
done
