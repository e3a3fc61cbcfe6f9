#!/bin/bash
# End-of-round evidence, round 4 (one gpurun call; most important first: the call may be cut by the remaining budget).
OUT=$PWD/gpurun_out/${1:-final_r04}; mkdir -p $OUT; REPO=$PWD
{ nproc; free -g | head -2; rocm-smi --showproductname 2>&1 | head -8; grep -m1 "model name" /proc/cpuinfo; } > $OUT/env.txt 2>&1
timeout 900 python -m pytest tests -m gpu -q --tb=short > $OUT/pytest_gpu.txt 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.txt; grep -E "passed|failed" $OUT/pytest_gpu.txt | tail -2
( time timeout 600 python bench.py ) > $OUT/bench_c2.json 2> $OUT/bench_c2.err; cut -c1-200 $OUT/bench_c2.json; tail -4 $OUT/bench_c2.err
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks_c2 -o ks -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu --pmc off --no-other --config 2 > $OUT/ks_c2.json 2> $OUT/ks_c2.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks_c5 -o ks -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu --pmc off --config 5 > $OUT/ks_c5.json 2> $OUT/ks_c5.err
run() { local name=$1; shift; timeout 300 rocprofv3 "$@" --output-format csv -d $OUT/prof/$name -o $name -- python $REPO/bench.py --steps 1 --warmup 1 --no-cpu --pmc off --no-other > $OUT/prof_$name.json 2> $OUT/prof_$name.err; }
run pmc_sq --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F64
run pmc_tcc --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU TCC_HIT_sum TCC_MISS_sum
run pmc_fetch --kernel-trace --pmc FETCH_SIZE
run pmc_write --kernel-trace --pmc WRITE_SIZE
run5() { local name=$1; shift; timeout 300 rocprofv3 "$@" --output-format csv -d $OUT/prof5/$name -o $name -- python $REPO/bench.py --steps 1 --warmup 1 --no-cpu --pmc off --config 5 > $OUT/prof5_$name.json 2> $OUT/prof5_$name.err; }
run5 pmc_sq --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY
run5 pmc_tcc --kernel-trace --pmc SQ_LDS_BANK_CONFLICT TCC_HIT_sum TCC_MISS_sum
run5 pmc_fetch --kernel-trace --pmc FETCH_SIZE
run5 pmc_write --kernel-trace --pmc WRITE_SIZE
cd $REPO
python scripts/pmc_summary.py $OUT/prof > $OUT/pmc_per_kernel.csv; grep -c "k_contract" $OUT/pmc_per_kernel.csv
python scripts/pmc_summary.py $OUT/prof5 > $OUT/pmc_per_kernel_c5.csv; grep -c "k_contract_sp" $OUT/pmc_per_kernel_c5.csv
for c in 5 3 4; do timeout 500 python bench.py --steps 3 --warmup 1 --config $c > $OUT/bench_c$c.json 2>> $OUT/bench.err; cut -c1-160 $OUT/bench_c$c.json; done
for k in 10 50 100; do timeout 300 python bench.py --steps 3 --warmup 1 --moving-window $k > $OUT/bench_mw$k.json 2>> $OUT/bench.err; cut -c1-160 $OUT/bench_mw$k.json; done
timeout 300 python scripts/mw_big_time.py --ldlt-only > $OUT/mw_big_time.txt 2>&1
timeout 200 python scripts/small_problem_latency.py > $OUT/small_problem_latency.txt 2>&1
MIK_FACTOR_CACHE=0 timeout 300 python scripts/sparse_time.py > $OUT/sparse_time.txt 2>&1
timeout 200 python scripts/pinv_block_time.py > $OUT/pinv_block_time.txt 2>&1
for g in 2 8; do timeout 400 python bench.py --gpus $g --steps 2 --warmup 1 --no-cpu > $OUT/bench_g$g.json 2>> $OUT/bench.err; cut -c1-120 $OUT/bench_g$g.json; done
timeout 400 python bench.py --gpus 8 --config 5 --steps 2 --warmup 1 --no-cpu > $OUT/bench_g8_c5.json 2>> $OUT/bench.err; cut -c1-120 $OUT/bench_g8_c5.json
timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu > $OUT/bench_2rank.json 2>> $OUT/bench.err; cut -c1-120 $OUT/bench_2rank.json
{ echo "MIK_FUZZ_CASES=${FUZZ:-3000} python -m pytest tests/test_randomized_parity.py -m gpu -q -s   (MI355X, HEAD of round 4: range-aware contraction on for spherical cases, drift equilibration, lane-per-point neighbour search, model-specialised LDL^T kernels)"; MIK_FUZZ_CASES=${FUZZ:-3000} timeout 600 python -m pytest tests/test_randomized_parity.py -m gpu -q -s 2>&1 | tail -4; } > $OUT/randomized.txt 2>&1; tail -3 $OUT/randomized.txt
rm -rf $OUT/prof/*/*.db $OUT/prof5/*/*.db $OUT/ks_c2/*.db $OUT/ks_c5/*.db 2>/dev/null
tail -2 $OUT/bench.err
