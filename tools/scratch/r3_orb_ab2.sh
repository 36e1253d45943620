cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "energy_statistics" 2>&1 | tail -3
grep -o '"energy_stat[a-z_A-Z]*": [-0-9.e]*' gpurun_out/parity_report_fullsize.json
bash tools/scratch/r3_orb_ab.sh lib_cur.so lib_unif.so
