# SQ counters of the headline step at HEAD (round 6: k_sweep_r8): instruction counts and where the waves wait.  bash tools/scratch/r6_sq_dmc.sh
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp; O=$R/gpurun_out/r6_sq_dmc; mkdir -p $O
run() { rocprofv3 --pmc $2 -d /tmp/sqd_$1 -o t -- python $R/bench.py --mode dmc --steps 4 --warmup 1 --no-cpu-baseline > /tmp/sqd_$1.log 2>&1 < /dev/null; tail -1 /tmp/sqd_$1.log | cut -c1-200; }
run a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES"
run b "SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU"
run c "SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM"
python - <<'PY' > $O/sq.txt
import sqlite3, os
d = {}
for s in "abc":
    p = f"/tmp/sqd_{s}/t_results.db"
    if not os.path.exists(p): continue
    c = sqlite3.connect(p)
    for k, n, v, cnt, dur in c.execute("select kernel_name, counter_name, avg(value), count(*), avg(duration) from counters_collection group by kernel_name, counter_name"):
        e = d.setdefault(k[:48], {}); e[n] = v; e["_n"] = cnt; e["_us"] = dur / 1e3
for k, v in sorted(d.items(), key=lambda kv: -kv[1].get("_n", 0) * kv[1].get("_us", 0)):
    if v["_n"] * v["_us"] < 1500: continue
    wc = v.get("SQ_WAVE_CYCLES", 1) or 1
    print(k.ljust(48), f"n={v['_n']} us={v['_us']:.1f}")
    print("    " + " ".join(f"{n[3:]}={x:.4g}" for n, x in sorted(v.items()) if n.startswith("SQ_")))
    print("    frac of wave cycles: wait_any %.2f wait_inst %.2f active %.2f | valu %.2f lds %.2f vmem %.2f sca %.2f | wait_inst_lds %.2f" % tuple(v.get(n, 0) / wc for n in
          ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA", "SQ_WAIT_INST_LDS")))
PY
cat $O/sq.txt
