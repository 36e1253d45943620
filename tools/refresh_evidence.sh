# Regenerates the measured evidence under gpurun_out/evidence at HEAD (copy what is to be judged into profiles/, named per round):
#   /usr/local/graft/bin/gpurun --timeout 3000 -- 'bash tools/refresh_evidence.sh'
# 1 bench line; 2 rocprofv3 kernel-trace summary of the same command; 3 PMC passes (FETCH_SIZE, WRITE_SIZE: separate runs, no
# trace options) of the bench at its default walker count + the counter calibration (tools/pmc_calib) -> pmc_summary.json;
# 4 the other BASELINE configurations; 5 their kernel summaries.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/evidence; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
W=${BENCH_WALKERS:-65536}
lscpu > $O/host_lscpu.txt
[ -x $R/tools/pmc_calib ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $R/tools/pmc_calib.hip -o $R/tools/pmc_calib 2>/dev/null  # (git-ignored binary: a fresh checkout has none)
(cd $R && python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt)
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d /tmp/pm_$c -o t -- python $R/bench.py --walkers $W --steps 2 --warmup 1 --settle 2 --no-cpu-baseline --no-profile --no-extra > /dev/null 2>&1 < /dev/null
  python $R/tools/pmc_counters.py /tmp/pm_$c/t_results.db $O/pmc_$c.csv
  rocprofv3 --pmc $c -d /tmp/pc_$c -o t -- $R/tools/pmc_calib > $O/pmc_calib_$c.txt 2>&1 < /dev/null
done
python $R/tools/pmc_summary.py /tmp/pm_FETCH_SIZE/t_results.db /tmp/pm_WRITE_SIZE/t_results.db /tmp/pc_FETCH_SIZE/t_results.db /tmp/pc_WRITE_SIZE/t_results.db $W $O/pmc_summary.json > $O/pmc_summary.txt 2>&1
cp $O/pmc_summary.json $R/profiles/r06_pmc_summary.json  # (this box's copy: bench.py reads its `traffic` fields from it)
python $R/bench.py --walkers $W > $O/bench.json 2> $O/bench.err < /dev/null
python $R/bench.py --mode dmc --steps 20 --warmup 2 > $O/bench_dmc.json 2>> $O/bench.err < /dev/null
python $R/bench.py --mode c4 --steps 20 --warmup 2 > $O/bench_c4.json 2>> $O/bench.err < /dev/null
rocprofv3 --kernel-trace --stats -d /tmp/pb -o b -- python $R/bench.py --walkers $W --steps 10 --warmup 2 --no-cpu-baseline --no-extra > /dev/null 2>&1 < /dev/null
python $R/tools/prof_stats.py /tmp/pb/b_results.db $O/bench_kernel_stats.csv
for case in k222 cubic; do  # counter passes of the periodic cases (separate runs, no trace options); calibration as above
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pkm_$c; rocprofv3 --pmc $c -d /tmp/pkm_$c -o t -- python $R/tools/pbc_bench.py --case $case --walkers 32768 --steps 1 --warmup 1 > /dev/null 2>&1 < /dev/null
    python $R/tools/pmc_counters.py /tmp/pkm_$c/t_results.db $O/pbc_${case}_pmc_$c.csv
  done
  python $R/tools/pmc_summary.py /tmp/pkm_FETCH_SIZE/t_results.db /tmp/pkm_WRITE_SIZE/t_results.db /tmp/pc_FETCH_SIZE/t_results.db /tmp/pc_WRITE_SIZE/t_results.db 32768 $O/pbc_${case}_pmc_summary.json > /dev/null 2>&1
  cp $O/pbc_${case}_pmc_summary.json $R/profiles/r06_pbc_${case}_pmc_summary.json
done
for c in k222 cubic; do for w in 8192 32768; do python $R/tools/pbc_bench.py --case $c --walkers $w --steps 4 2>/dev/null | tail -1 >> $O/pbc_bench.jsonl; done; done
rocprofv3 --kernel-trace --stats -d /tmp/pk -o k -- python $R/tools/pbc_bench.py --case k222 --walkers 32768 --steps 3 > /dev/null 2>&1 < /dev/null
python $R/tools/prof_stats.py /tmp/pk/k_results.db $O/pbc_k222_kernel_stats.csv
python $R/tools/config_bench.py c2 --walkers 4096 --steps 20 2>/dev/null | tail -1 >> $O/config_bench.jsonl
python $R/tools/config_bench.py c2 --walkers 65536 --steps 20 2>/dev/null | tail -1 >> $O/config_bench.jsonl
python $R/tools/config_bench.py c3 --walkers 4096 --steps 8 2>/dev/null | tail -1 >> $O/config_bench.jsonl
python $R/tools/config_bench.py c3 --walkers 8192 --steps 8 2>/dev/null | tail -1 >> $O/config_bench.jsonl
python $R/tools/config_bench.py c3 --walkers 32768 --steps 8 2>/dev/null | tail -1 >> $O/config_bench.jsonl
python $R/tools/config_bench.py c4 --walkers 2048 --steps 20 2>/dev/null | tail -1 >> $O/config_bench.jsonl
python $R/tools/config_bench.py c4 --walkers 16384 --steps 20 2>/dev/null | tail -1 >> $O/config_bench.jsonl
python $R/tools/config_bench.py c5 --walkers 4096 --steps 10 2>/dev/null | tail -1 >> $O/config_bench.jsonl
python $R/tools/config_bench.py c5 --walkers 16384 --steps 10 2>/dev/null | tail -1 >> $O/config_bench.jsonl
python $R/tools/config_bench.py c5 --walkers 32768 --steps 10 2>/dev/null | tail -1 >> $O/config_bench.jsonl
rocprofv3 --kernel-trace --stats -d /tmp/pd -o d -- python $R/tools/config_bench.py c5 --walkers 16384 --steps 10 > /dev/null 2>&1 < /dev/null
python $R/tools/prof_stats.py /tmp/pd/d_results.db $O/dmc_c5_kernel_stats.csv
rm -rf /tmp/pd; rocprofv3 --kernel-trace --stats -d /tmp/pd -o d -- python $R/tools/config_bench.py c5 --walkers 4096 --steps 10 > /dev/null 2>&1 < /dev/null
python $R/tools/prof_stats.py /tmp/pd/d_results.db $O/dmc_c5_4096_kernel_stats.csv
rm -rf /tmp/pd; rocprofv3 --kernel-trace --stats -d /tmp/pd -o d -- python $R/tools/config_bench.py c4 --walkers 2048 --steps 4 > /dev/null 2>&1 < /dev/null
python $R/tools/prof_stats.py /tmp/pd/d_results.db $O/c4_2048_kernel_stats.csv
for w in 1024 2048 4096 8192 16384 32768; do echo -n "{\"walkers\": $w, \"line\": \"" >> $O/small_shards.txt; python $R/tools/scratch/lib_bench.py $R/pyqmc_amd/lib/libpyqmc_amd.so $w 2>/dev/null | tr -d '\n' >> $O/small_shards.txt; echo "\"}" >> $O/small_shards.txt; done
rm -rf /tmp/pd; rocprofv3 --kernel-trace --stats -d /tmp/pd -o d -- python $R/tools/scratch/lib_bench.py $R/pyqmc_amd/lib/libpyqmc_amd.so 4096 > /dev/null 2>&1 < /dev/null
python $R/tools/prof_stats.py /tmp/pd/d_results.db $O/m_4096_kernel_stats.csv
# round 4: the pipelined half-ensemble sweep (A/B per mode + which kernels really overlap), the 2-rank same-GPU DMC rehearsal
python $R/tools/split_ab.py --walkers $W --steps 10 1 2 3 > $O/split_ab.jsonl 2>> $O/bench.err < /dev/null
for m in 0 1 2; do rm -rf /tmp/tr$m; rocprofv3 --kernel-trace --output-format csv -d /tmp/tr$m -o t -- python $R/tools/split_ab.py --walkers $W --steps 3 $m > /dev/null 2>&1 < /dev/null; echo "== PQA_SPLIT=$m (first block: mode 0 reference run, second: the mode)"; python $R/tools/trace_overlap.py $(find /tmp/tr$m -name "*kernel_trace.csv" | head -1) 0.25; done > $O/split_overlap.txt 2>&1
(cd $R && python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --mode dmc --backend gloo --same-gpu --device-buffers --unbalance 0.5 --steps 20 --warmup 1 > $O/bench_dmc_2rank_same_gpu.json 2>> $O/bench.err < /dev/null)
python $R/tools/cpu_config_baseline.py c2 c3 c4 c5 big > $O/cpu_config_baseline.jsonl 2>> $O/bench.err
# round 5: handles beyond 64 electrons per spin, the protocol route, the resident sweep against the launch-per-move sweep, the LDS-DMA probe
for c in "big --walkers 1024" "big --walkers 8192" "big_pbc --walkers 256" "big_complex --walkers 128"; do python $R/tools/config_bench.py $c --steps 1 2>/dev/null | tail -1 >> $O/config_bench.jsonl; done
for w in 4096 65536; do python $R/tools/protocol_profile.py $w --json $O/protocol_$w.json > /dev/null 2>> $O/bench.err; done
for w in 1024 4096 16384 65536; do for r in 1 0; do echo -n "{\"walkers\": $w, \"PQA_RES\": $r, \"line\": \"" >> $O/resident_ab.txt; PQA_RES=$r python $R/tools/scratch/lib_bench.py $R/pyqmc_amd/lib/libpyqmc_amd.so $w 2>/dev/null | tr -d '\n' >> $O/resident_ab.txt; echo "\"}" >> $O/resident_ab.txt; done; done
rm -rf /tmp/pd; rocprofv3 --kernel-trace --stats -d /tmp/pd -o d -- python $R/tools/scratch/lib_bench.py $R/pyqmc_amd/lib/libpyqmc_amd.so 4096 > /dev/null 2>&1 < /dev/null
python $R/tools/prof_stats.py /tmp/pd/d_results.db $O/m_4096_kernel_stats.csv
[ -x $R/tools/scratch/bin/dma_probe ] || (mkdir -p $R/tools/scratch/bin && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 $R/tools/scratch/dma_probe.hip -o $R/tools/scratch/bin/dma_probe 2>/dev/null)
timeout 120 $R/tools/scratch/bin/dma_probe > $O/dma_probe.txt 2>&1
# round 5, later sessions: the periodic / complex resident sweep and the one-launch wave-per-walker sweep against the launches they replace, the
# ECP point totals left on the device against the read-back
for c in c3 c5; do for w in 4096 8192 16384; do for r in 1 0; do echo -n "{\"PQA_RES\": $r, \"line\": " >> $O/resident_pbc_ab.txt; PQA_RES=$r python $R/tools/config_bench.py $c --walkers $w --steps 8 2>/dev/null | tail -1 | tr -d '\n' >> $O/resident_pbc_ab.txt; echo "}" >> $O/resident_pbc_ab.txt; done; done; done
for w in 1024 2048 4096; do for m in "PQA_WW=0 PQA_ECP_DEFER=0 PQA_EN_OVERLAP=0" "PQA_WW=0" "PQA_WW=3" "PQA_WW=1 PQA_ECP_DEFER=0" "PQA_WW=1"; do echo -n "{\"env\": \"$m\", \"line\": " >> $O/c4_one_launch_ab.txt; env $m python $R/tools/config_bench.py c4 --walkers $w --steps 20 2>/dev/null | tail -1 | tr -d '\n' >> $O/c4_one_launch_ab.txt; echo "}" >> $O/c4_one_launch_ab.txt; done; done
for d in 0 1; do echo -n "{\"PQA_ECP_DEFER\": $d, \"line\": " >> $O/c4_one_launch_ab.txt; PQA_ECP_DEFER=$d python $R/tools/config_bench.py c2 --walkers 4096 --steps 40 2>/dev/null | tail -1 | tr -d '\n' >> $O/c4_one_launch_ab.txt; echo "}" >> $O/c4_one_launch_ab.txt; done
# round 6: the second-generation resident sweep (k_sweep_r8) against k_sweep_res and the launch-per-move sweep, its phase stamps (timing build
# libpqa_RCLK.so, in-tree: python -c "import __graft_entry__ as g, os; g.build(extra_flags=['-DPQA_RES_CLK'], lib=os.path.join(g.LIBDIR, 'libpqa_RCLK.so'))"),
# the MFMA / FMA issue-rate probe behind its schedule, and the N = 1 point of the strong-scaling curve (65 536 walkers in all)
python $R/tools/scratch/r8_scan.py M 2048 4096 8192 16384 32768 65536 > $O/resident_r8_ab.jsonl 2>> $O/bench.err < /dev/null
python $R/tools/scratch/r8_scan.py C2 1024 4096 16384 >> $O/resident_r8_ab.jsonl 2>> $O/bench.err < /dev/null
if [ -f $R/pyqmc_amd/lib/libpqa_RCLK.so ]; then for w in 2048 4096 16384; do PQA_LIB=$R/pyqmc_amd/lib/libpqa_RCLK.so PQA_RES=1 python $R/tools/scratch/res_clk.py $w; done > $O/r8_phase_stamps.txt 2>&1; fi
[ -x $R/tools/scratch/bin/mfma_probe ] || (mkdir -p $R/tools/scratch/bin && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-result $R/tools/scratch/mfma_probe.hip -o $R/tools/scratch/bin/mfma_probe 2>/dev/null)
timeout 120 $R/tools/scratch/bin/mfma_probe > $O/mfma_probe.txt 2>&1
python $R/bench.py --scaling strong --walkers $W --no-cpu-baseline --no-extra > $O/bench_strong_n1.json 2>> $O/bench.err < /dev/null
# round 6, energy pass: what the row cache and the inverse planes stream at alone and mixed (k_kinetic_lw's bound); SQ counters of the headline step
# and of the C5 DMC step (instruction counts, pipe busy, where the waves wait)
[ -x $R/tools/scratch/bin/row_probe ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-result $R/tools/scratch/row_probe.hip -o $R/tools/scratch/bin/row_probe 2>/dev/null
timeout 120 $R/tools/scratch/bin/row_probe > $O/row_probe.txt 2>&1
timeout 600 bash $R/tools/scratch/r6_sq.sh > /dev/null 2>&1; cp $R/gpurun_out/r6_sq/sq.txt $O/sq_counters.txt 2>/dev/null
timeout 600 bash $R/tools/scratch/r6_sq_dmc.sh > /dev/null 2>&1; cp $R/gpurun_out/r6_sq_dmc/sq.txt $O/sq_counters_dmc.txt 2>/dev/null
cp $R/gpurun_out/parity_report.json $R/gpurun_out/parity_report_fullsize.json $O/ 2>/dev/null
ls -la $O
