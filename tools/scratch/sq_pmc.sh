# SQ counters of the sweep's kernels: where do the waves spend their cycles?  bash tools/scratch/sq_pmc.sh
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD -d /tmp/sq -o t -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-extra > /dev/null 2>&1 < /dev/null
python - <<'PY'
import sqlite3
c = sqlite3.connect("/tmp/sq/t_results.db")
rows = c.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection group by kernel_name, counter_name").fetchall()
d = {}
for k, n, v, cnt in rows:
    d.setdefault(k[:60], {})[n] = v
for k, v in d.items():
    if any(s in k for s in ("k_move_part", "k_commit", "k_flush", "k_orb<5", "k_kinetic", "k_accept_fin", "k_transpose")):
        wc = v.get("SQ_WAVE_CYCLES", 1)
        print(k[:44].ljust(44), " ".join(f"{n[3:]}={v.get(n,0):.3g}" for n in ("SQ_WAVE_CYCLES","SQ_BUSY_CYCLES","SQ_INSTS_VALU","SQ_INSTS_VMEM_RD")),
              "| wait_any %.2f wait_inst %.2f active %.2f valu %.2f" % (v.get("SQ_WAIT_ANY",0)/wc, v.get("SQ_WAIT_INST_ANY",0)/wc, v.get("SQ_ACTIVE_INST_ANY",0)/wc, v.get("SQ_ACTIVE_INST_VALU",0)/wc))
PY
