R=$GRAFT_REPO_ROOT; cd $R
run() { echo -n "$* : "; env "$@" python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-extra 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), round(d['value']), round(d['roofline']['frac'],4), d['roofline']['launches'], round(d['roofline_hbm']['avg_launch_ms'],4), d['roofline_hbm']['launches'], round(d['roofline_hbm_flush']['avg_launch_ms'],4), d['roofline_hbm_flush']['launches'])"; }
run PQA_PROF_STRIDE=4
run PQA_PROF_STRIDE=16
run PQA_PROF_STRIDE=32
run PQA_PROF_STRIDE=4
run PQA_PROF_STRIDE=16
