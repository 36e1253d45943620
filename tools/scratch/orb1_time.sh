#!/bin/bash
# k_orb<1> time per launch in the bench for given libraries: bash tools/scratch/orb1_time.sh lib.so ...
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  rm -rf /tmp/o1prof
  PQA_LIB=$lib timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/o1prof -o o1 -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extra > /tmp/o1.log 2>&1 < /dev/null
  echo "== $lib"
  timeout 60 python $R/tools/prof_stats.py /tmp/o1prof/o1_results.db /tmp/o1.csv < /dev/null > /dev/null 2>&1
  grep -E "k_orb<1|k_ecp|k_kinetic_lw" /tmp/o1.csv < /dev/null | cut -c1-50,100-300
done
