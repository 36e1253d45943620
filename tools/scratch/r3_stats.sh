cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r3_stats; mkdir -p $O; rm -f $O/*
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
grep -o '"energy_stat[a-z_A-Z]*": [-0-9.e]*' gpurun_out/parity_report_fullsize.json
timeout 1500 python tools/cpu_config_baseline.py c2 c3 c4 c5 > $O/cpu_config_baseline.jsonl 2> $O/cpu.err; cut -c1-330 $O/cpu_config_baseline.jsonl; tail -3 $O/cpu.err
