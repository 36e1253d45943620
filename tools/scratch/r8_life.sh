#!/bin/bash
# Block lifetime of k_sweep_r8 (timing build libpqa_RCLK.so) at the given walker counts, and the resident-vs-launches parity test on the same library
export PQA_LIB=${PQA_LIB:-pyqmc_amd/lib/libpqa_RCLK.so} PQA_RES=1
for W in "$@"; do
  echo -n "W $W: "; timeout 120 python tools/scratch/res_clk.py $W 2>&1 | grep "block lifetime"
done
