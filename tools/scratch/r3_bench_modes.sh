cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r3_modes; mkdir -p $O; rm -f $O/*
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 3000 $O/bench.json
timeout 600 python bench.py --mode dmc --steps 10 --warmup 1 > $O/bench_dmc.json 2> $O/bench_dmc.err; cat $O/bench_dmc.json | cut -c1-1500
timeout 600 python bench.py --mode c4 --steps 10 --warmup 2 > $O/bench_c4.json 2> $O/bench_c4.err; cat $O/bench_c4.json | cut -c1-1500
for m in "--mode vmc --walkers 8192 --steps 4 --warmup 1 --settle 2 --no-cpu-baseline --no-extra" "--mode dmc --scaling strong --walkers 4096 --steps 5 --warmup 1" "--mode c4 --scaling strong --walkers 4096 --steps 4"; do
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --backend gloo --same-gpu $m 2>> $O/mp.err | tail -1 | cut -c1-1800 | tee -a $O/mp.jsonl
done
tail -5 $O/mp.err
