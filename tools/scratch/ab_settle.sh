cd $GRAFT_REPO_ROOT
f() { python - <<'PY'
import json,sys
for l in open('/tmp/o.json'):
    if l.startswith('{'):
        d=json.loads(l); print(round(d['value']), round(d['ms_per_step'],2), round(d['extra']['sweep_plus_energy']['ms_per_step'],2) if 'extra' in d else None)
PY
}
for i in 1 2; do for s in 3 30 90 200; do
echo settle $s; timeout 300 python bench.py --no-cpu-baseline --settle $s > /tmp/o.json 2>/dev/null; f
done; done
