#!/bin/bash
# kernel times of the C5 DMC bench under env settings: bash tools/scratch/dmc_time.sh "ENV=1 ..." ...
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for envs in "$@"; do
  rm -rf /tmp/o2prof
  env $envs timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/o2prof -o o2 -- python $R/bench.py --mode dmc --steps 20 --warmup 2 --no-cpu-baseline --no-extra > /tmp/o2.log 2>&1 < /dev/null
  echo "== $envs"; grep '^{' /tmp/o2.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" || tail -3 /tmp/o2.log
  timeout 60 python $R/tools/prof_stats.py /tmp/o2prof/o2_results.db /tmp/o2.csv < /dev/null > /dev/null 2>&1
  python - <<'PY'
import csv
rows = list(csv.reader(open('/tmp/o2.csv')))[1:]
tot = sum(float(r[2]) for r in rows)
print('total_us', tot)
for r in rows[:12]: print(r[0][:44].ljust(44), r[1], r[2], r[3])
PY
done
