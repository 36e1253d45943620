#!/bin/bash
# Block lifetime of k_sweep_r8 with phases left out (timing build libpqa_RCLK.so, PQA_R8_ABL bit mask): bash tools/scratch/r8_abl.sh W...
export PQA_LIB=pyqmc_amd/lib/libpqa_RCLK.so PQA_RES=1
for W in "$@"; do
  for a in 0 1 2 4 8 16 3 7 31; do
    echo -n "W $W abl $a: "; PQA_R8_ABL=$a timeout 120 python tools/scratch/res_clk.py $W 2>&1 | grep "block lifetime"
  done
done
