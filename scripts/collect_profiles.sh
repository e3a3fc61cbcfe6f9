#!/bin/bash
# Copy the summaries of an evidence run (gpurun_out/TAG, written by scripts/gpu_final.sh) into profiles/ (tracked).
TAG=${1:?tag}; R=${2:-r03}; S=gpurun_out/$TAG; P=profiles
cp() { [ -e "$1" ] && command cp "$1" "$2"; }   # a partial run leaves some files out: keep the older summaries
for c in 2 3 4 5; do cp $S/bench_c$c.json $P/${R}_bench_c$c.json; cp $S/ks_c$c/ks_kernel_stats.csv $P/${R}_bench_c${c}_rocprofv3_kernel_stats.csv; done
for k in 10 50 100; do cp $S/bench_mw$k.json $P/${R}_bench_moving_window_k$k.json; done
cp $S/ks_mw50/ks_kernel_stats.csv $P/${R}_bench_moving_window_k50_rocprofv3_kernel_stats.csv
cp $S/bench_g2.json $P/${R}_bench_c2_group2_aliased_on_1gpu.json; cp $S/bench_g4.json $P/${R}_bench_c2_group4_aliased_on_1gpu.json
cp $S/bench_g8.json $P/${R}_bench_c2_group8_aliased_on_1gpu.json; cp $S/bench_g8_c5.json $P/${R}_bench_c5_group8_aliased_on_1gpu.json
cp $S/bench_2rank.json $P/${R}_bench_torchrun_2ranks_on_1gpu.json
cp $S/pytest_gpu.txt $P/${R}_pytest_gpu.txt; cp $S/env.txt $P/${R}_env.txt; cp $S/stat_time.txt $P/${R}_statistics_timing.txt
cp $S/small_problem_latency.txt $P/${R}_small_problem_latency.txt
cp $S/inverse_ab.txt $P/${R}_inverse_variants_ab.txt; cp $S/execute_overhead.txt $P/${R}_execute_overhead.txt
cp $S/reference_benchmark_shapes.txt $P/${R}_reference_benchmark_shapes.txt; cp $S/pmc_per_kernel.csv $P/${R}_bench_c2_rocprofv3_pmc_per_kernel.csv
cp $S/inverse_timeline_final.txt $P/${R}_inverse_timeline_final_state.txt
cp $S/randomized_3000.txt $P/${R}_randomized_parity_3000_cases.txt; cp $S/execute_breakdown.txt $P/${R}_execute_breakdown.txt
python scripts/make_traffic_json.py $P/${R}_bench_c2_rocprofv3_pmc_per_kernel.csv $P/${R}_bench_c2.json > $P/k_contract_traffic.json.tmp && mv $P/k_contract_traffic.json.tmp $P/k_contract_traffic.json
