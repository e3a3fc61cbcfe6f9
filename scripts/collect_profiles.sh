#!/bin/bash
# Copy the summaries of an evidence run (gpurun_out/TAG, written by scripts/gpu_final.sh) into profiles/ (tracked).
TAG=${1:?tag}; R=${2:-r01}; S=gpurun_out/$TAG; P=profiles
cp $S/bench.json $P/${R}_bench_c2.json
for c in 3 4 5; do cp $S/bench_c$c.json $P/${R}_bench_c$c.json; done
for k in 10 50; do cp $S/bench_mw$k.json $P/${R}_bench_moving_window_k$k.json; done
cp $S/bench_2rank.json $P/${R}_bench_torchrun_2ranks_on_1gpu.json
cp $S/pytest_gpu.txt $P/${R}_pytest_gpu.txt
cp $S/kernel_bench.txt $P/${R}_kernel_bench.txt
cp $S/ubench_f64.txt $P/${R}_ubench_f64.txt
cp $S/stat_time.txt $P/${R}_statistics_timing.txt
cp $S/mw_big_time.txt $P/${R}_moving_window_timing.txt
cp $S/execute_overhead.txt $P/${R}_execute_overhead.txt
cp $S/prof/ktrace/ktrace_kernel_stats.csv $P/${R}_bench_c2_rocprofv3_kernel_stats.csv
python scripts/pmc_summary.py $S/prof > $P/${R}_bench_c2_rocprofv3_pmc_per_kernel.csv
cp $S/prof_mw/mw_kernel_stats.csv $P/${R}_bench_moving_window_k50_rocprofv3_kernel_stats.csv 2>/dev/null || true
