R=$GRAFT_REPO_ROOT; cd $R
run() { echo -n "$* : "; python bench.py --no-cpu-baseline --no-extra "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), round(d['value']), d['ranks'][0]['pci_bus_id'])"; }
run --steps 60
run --steps 60 --no-profile
run --steps 30 --warmup 3 --no-profile
run --steps 60
run --steps 60 --no-profile
rocm-smi --showclocks 2>/dev/null | head -20
