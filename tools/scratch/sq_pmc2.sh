# second counter set: scalar side of the sweep's kernels
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE -d /tmp/sq2 -o t -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-extra > /tmp/sq2.log 2>&1 < /dev/null
tail -2 /tmp/sq2.log
python - <<'PY'
import sqlite3
c = sqlite3.connect("/tmp/sq2/t_results.db")
rows = c.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection group by kernel_name, counter_name").fetchall()
d = {}
for k, n, v, cnt in rows:
    d.setdefault(k[:60], {})[n] = v
for k, v in d.items():
    if any(s in k for s in ("k_move_part", "k_commit", "k_flush", "k_orb<5", "k_kinetic", "k_accept_fin")):
        print(k[:44].ljust(44), " ".join(f"{n}={x:.4g}" for n, x in sorted(v.items())))
PY
