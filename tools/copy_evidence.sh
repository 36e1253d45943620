# copies gpurun_out/evidence/* (tools/refresh_evidence.sh) into profiles/ under this round's names; run from the repository root
set -e
R=$(cd "$(dirname "$0")/.." && pwd); E=$R/gpurun_out/evidence; P=$R/profiles; N=${1:-r06}
test -f $E/bench.json
for f in bench.json bench_c4.json bench_dmc.json bench_kernel_stats.csv config_bench.jsonl cpu_config_baseline.jsonl host_lscpu.txt parity_report.json parity_report_fullsize.json pbc_bench.jsonl pbc_k222_kernel_stats.csv pbc_k222_pmc_summary.json pbc_cubic_pmc_summary.json pmc_summary.json dmc_c5_kernel_stats.csv dmc_c5_4096_kernel_stats.csv c4_2048_kernel_stats.csv m_4096_kernel_stats.csv small_shards.txt split_ab.jsonl split_overlap.txt bench_dmc_2rank_same_gpu.json protocol_4096.json protocol_65536.json resident_ab.txt dma_probe.txt resident_pbc_ab.txt c4_one_launch_ab.txt pytest_gpu.txt resident_r8_ab.jsonl r8_phase_stamps.txt mfma_probe.txt bench_strong_n1.json row_probe.txt sq_counters.txt sq_counters_dmc.txt; do cp $E/$f $P/${N}_$f; done
cp $E/pmc_FETCH_SIZE.csv $P/${N}_pmc_fetch_size.csv; cp $E/pmc_WRITE_SIZE.csv $P/${N}_pmc_write_size.csv
cp $E/pbc_k222_pmc_FETCH_SIZE.csv $P/${N}_pbc_k222_pmc_fetch_size.csv; cp $E/pbc_k222_pmc_WRITE_SIZE.csv $P/${N}_pbc_k222_pmc_write_size.csv
cat $E/pmc_calib_FETCH_SIZE.txt $E/pmc_calib_WRITE_SIZE.txt > $P/${N}_pmc_calib.txt
echo copied
