#!/bin/bash
# kernel times of the energy pass in the bench under env settings: bash tools/scratch/kin_time.sh "ENV=1 ..." ...
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for envs in "$@"; do
  rm -rf /tmp/o1prof
  env $envs timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/o1prof -o o1 -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extra > /tmp/o1.log 2>&1 < /dev/null
  echo "== $envs"
  timeout 60 python $R/tools/prof_stats.py /tmp/o1prof/o1_results.db /tmp/o1.csv < /dev/null > /dev/null 2>&1
  python - <<'PY'
import csv
for r in csv.reader(open('/tmp/o1.csv')):
    if r and any(k in r[0] for k in ('k_kinetic','k_jas_sym','k_orb<1','k_ecp_point')): print(r[0][:40], r[1:])
PY
done
