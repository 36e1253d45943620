# where do the 1.1 ms between bench.py's timed step and extra.sweep_plus_energy go?  (event brackets / process warm-up)
cd $GRAFT_REPO_ROOT
f() { python - <<'PY'
import json,sys
for l in open('/tmp/o.json'):
    if l.startswith('{'):
        d=json.loads(l); print(round(d['value']), round(d['ms_per_step'],2), round(d['extra']['sweep_plus_energy']['ms_per_step'],2) if 'extra' in d else None)
PY
}
for i in 1 2; do
echo default; timeout 300 python bench.py --no-cpu-baseline > /tmp/o.json 2>/dev/null; f
echo noprofile; timeout 300 python bench.py --no-cpu-baseline --no-profile > /tmp/o.json 2>/dev/null; f
echo stride64; PQA_PROF_STRIDE=64 timeout 300 python bench.py --no-cpu-baseline > /tmp/o.json 2>/dev/null; f
echo steps24; timeout 300 python bench.py --no-cpu-baseline --steps 24 --warmup 8 > /tmp/o.json 2>/dev/null; f
done
