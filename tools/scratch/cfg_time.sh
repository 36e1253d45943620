#!/bin/bash
# kernel times of a tools/config_bench.py configuration: bash tools/scratch/cfg_time.sh big 8192 [steps]
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/o3prof
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/o3prof -o o3 -- python $R/tools/config_bench.py $1 --walkers $2 --steps ${3:-4} > /tmp/o3.log 2>&1 < /dev/null
grep '^{' /tmp/o3.log | tail -1 | cut -c1-300
timeout 60 python $R/tools/prof_stats.py /tmp/o3prof/o3_results.db /tmp/o3.csv < /dev/null > /dev/null 2>&1
python - <<'PY'
import csv
rows = list(csv.reader(open('/tmp/o3.csv')))[1:]
tot = sum(float(r[2]) for r in rows)
print('total_us', tot)
for r in rows[:16]: print(r[0][:64].ljust(64), r[1], r[2], r[3], r[4])
PY
