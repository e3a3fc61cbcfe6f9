#!/bin/bash
# End-of-round evidence run (round 3): GPU tests, all bench configs with the full CPU protocol, moving-window bench lines,
# device-group and launcher runs on the one GPU (aliased), rocprofv3 kernel stats + PMC of the default schedule, timing scripts.
OUT=$PWD/gpurun_out/${1:-final}; mkdir -p $OUT; REPO=$PWD
{ nproc; free -g | head -2; rocm-smi --showproductname 2>&1 | head -8; grep -m1 "model name" /proc/cpuinfo; } > $OUT/env.txt 2>&1
timeout 1200 python -m pytest tests -m gpu -q --tb=short -s > $OUT/pytest_gpu.txt 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.txt; tail -3 $OUT/pytest_gpu.txt
{ echo "MIK_FUZZ_CASES=3000 python -m pytest tests/test_randomized_parity.py -m gpu -q -s   (MI355X, HEAD of round 3: probe-verified inverse, device-generated grids, zero-copy results, new moving-window kernels)"; MIK_FUZZ_CASES=3000 timeout 900 python -m pytest tests/test_randomized_parity.py -m gpu -q -s 2>&1 | tail -6; } > $OUT/randomized_3000.txt 2>&1; tail -2 $OUT/randomized_3000.txt
timeout 900 python bench.py --steps 5 --warmup 2 > $OUT/bench_c2.json 2> $OUT/bench.err; cut -c1-300 $OUT/bench_c2.json
for c in 3 4 5; do timeout 900 python bench.py --steps 3 --warmup 1 --config $c > $OUT/bench_c$c.json 2>> $OUT/bench.err; done
for k in 10 50 100; do timeout 600 python bench.py --steps 3 --warmup 1 --moving-window $k > $OUT/bench_mw$k.json 2>> $OUT/bench.err; done
timeout 600 python bench.py --gpus 2 --steps 3 --warmup 1 > $OUT/bench_g2.json 2>> $OUT/bench.err
timeout 600 python bench.py --gpus 4 --steps 2 --warmup 1 > $OUT/bench_g4.json 2>> $OUT/bench.err
timeout 600 python bench.py --gpus 8 --steps 1 --warmup 1 > $OUT/bench_g8.json 2>> $OUT/bench.err
timeout 900 python bench.py --gpus 8 --config 5 --steps 1 --warmup 1 > $OUT/bench_g8_c5.json 2>> $OUT/bench.err
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --steps 2 --warmup 1 > $OUT/bench_2rank.json 2> $OUT/bench_2rank.err
timeout 300 python scripts/stat_time.py > $OUT/stat_time.txt 2>&1
timeout 300 python scripts/execute_breakdown.py 2 3 4 > $OUT/execute_breakdown.txt 2>&1
timeout 300 python scripts/small_problem_latency.py > $OUT/small_problem_latency.txt 2>&1
timeout 300 python scripts/inverse_lookahead_ab.py > $OUT/inverse_ab.txt 2>&1
{ for g in 1 0; do echo "MIK_DEVICE_GRID=$g"; MIK_DEVICE_GRID=$g timeout 300 python scripts/execute_overhead.py 2 2>&1 | head -2; done; } > $OUT/execute_overhead.txt
timeout 600 python scripts/reference_benchmark_shapes.py > $OUT/reference_benchmark_shapes.txt 2>&1
cd /tmp && export TMPDIR=/tmp
for c in 2 3 4 5; do timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks_c$c -o ks -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu --pmc off --config $c > $OUT/ks_c$c.json 2> $OUT/ks_c$c.err; done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks_mw50 -o ks -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu --pmc off --moving-window 50 > $OUT/ks_mw50.json 2> $OUT/ks_mw50.err
# PMC of the DEFAULT (event-ordered) schedule, one counter group per pass (gpurun refuses --pmc next to other trace domains)
run() { local name=$1; shift; timeout 600 rocprofv3 "$@" --output-format csv -d $OUT/prof/$name -o $name -- python $REPO/bench.py --steps 1 --warmup 1 --no-cpu --pmc off > $OUT/prof_$name.json 2> $OUT/prof_$name.err; }
run pmc_sq --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F64
run pmc_tcc --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU TCC_HIT_sum TCC_MISS_sum
run pmc_fetch --kernel-trace --pmc FETCH_SIZE
run pmc_write --kernel-trace --pmc WRITE_SIZE
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/tl_5000 -o tl -- python $REPO/scripts/inverse_timeline.py run 5000 > $OUT/tl_5000.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/tl_8000 -o tl -- python $REPO/scripts/inverse_timeline.py run 8000 > $OUT/tl_8000.txt 2>&1
cd $REPO
python scripts/pmc_summary.py $OUT/prof > $OUT/pmc_per_kernel.csv
{ echo "== N=5000 (defaults)"; grep invert_ms $OUT/tl_5000.txt; python scripts/inverse_timeline.py parse $OUT/tl_5000; echo "== N=8000 (defaults)"; grep invert_ms $OUT/tl_8000.txt; python scripts/inverse_timeline.py parse $OUT/tl_8000; } > $OUT/inverse_timeline_final.txt 2>&1
grep -c "k_contract" $OUT/pmc_per_kernel.csv; tail -2 $OUT/bench.err
