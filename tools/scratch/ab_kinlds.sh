# k_kinetic_lwx (coordinates in LDS) against k_kinetic_lw: PQA_KIN_LDS=1/0
cd $GRAFT_REPO_ROOT
f() { python - <<'PY'
import json
for l in open('/tmp/o.json'):
    if l.startswith('{'):
        d=json.loads(l); print(round(d['value']), round(d['ms_per_step'],2), {k:round(v['ms_per_step'],2) for k,v in d['extra']['by_walkers_per_gpu'].items()})
PY
}
for i in 1 2; do for k in 0 1; do
echo kin_lds $k; PQA_KIN_LDS=$k timeout 300 python bench.py --no-cpu-baseline > /tmp/o.json 2>/dev/null; f
done; done
for k in 0 1; do
for c in "c5 4096" "c5 16384" "c3 8192" "c2 4096" "c4 2048"; do echo kin_lds $k $c; PQA_KIN_LDS=$k timeout 300 python tools/config_bench.py $c --steps 8 2>/dev/null | tail -1 | cut -c1-160; done
done
