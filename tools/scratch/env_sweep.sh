# headline ms/step under env switches: bash tools/scratch/env_sweep.sh
R=$GRAFT_REPO_ROOT; cd $R
run() { echo -n "$* : "; env "$@" python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-extra --no-profile 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), round(d['value']))"; }
run PQA_NONE=1
run PQA_LW_KB=4
run PQA_LW_KB=6
run PQA_LW_KB=8
run PQA_LW_GM=2
run PQA_LW_GM=8
run PQA_NONE=1
