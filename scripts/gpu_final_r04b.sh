#!/bin/bash
# End-of-round evidence, round 4, second session (one gpurun call; most important first).  What the first session's run
# (scripts/gpu_final_r04.sh, profiles/r04_*) measured and this session did not touch -- moving-window lines k = 50 / 100, pseudo-inverse,
# small-problem latency, HBM split of the dense contraction -- is not repeated.
OUT=$PWD/gpurun_out/${1:-final_r04b}; mkdir -p $OUT; REPO=$PWD
{ nproc; free -g | head -2; rocm-smi --showproductname 2>&1 | head -8; grep -m1 "model name" /proc/cpuinfo; } > $OUT/env.txt 2>&1
timeout 900 python -m pytest tests -m gpu -q --tb=short > $OUT/pytest_gpu.txt 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.txt; grep -E "passed|failed" $OUT/pytest_gpu.txt | tail -2
( time timeout 600 python bench.py ) > $OUT/bench_c2.json 2> $OUT/bench_c2.err; cut -c1-200 $OUT/bench_c2.json; tail -4 $OUT/bench_c2.err
timeout 500 python bench.py --steps 3 --warmup 1 --config 5 > $OUT/bench_c5.json 2>> $OUT/bench.err; cut -c1-160 $OUT/bench_c5.json
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks_c5 -o ks -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu --pmc off --config 5 > $OUT/ks_c5.json 2> $OUT/ks_c5.err
run5() { local name=$1; shift; timeout 300 rocprofv3 "$@" --output-format csv -d $OUT/prof5/$name -o $name -- python $REPO/bench.py --steps 1 --warmup 1 --no-cpu --pmc off --config 5 > $OUT/prof5_$name.json 2> $OUT/prof5_$name.err; }
run5 pmc_sq --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY
run5 pmc_tcc --kernel-trace --pmc SQ_LDS_BANK_CONFLICT TCC_HIT_sum TCC_MISS_sum
run5 pmc_fetch --kernel-trace --pmc FETCH_SIZE
run5 pmc_write --kernel-trace --pmc WRITE_SIZE
cd $REPO
python scripts/pmc_summary.py $OUT/prof5 > $OUT/pmc_per_kernel_c5.csv; grep -c "k_contract_spg" $OUT/pmc_per_kernel_c5.csv
timeout 300 python scripts/sparse_rows_ab.py --dense > $OUT/sparse_rows_ab.txt 2>&1; tail -3 $OUT/sparse_rows_ab.txt
timeout 200 python scripts/mw_sorted_ab.py > $OUT/mw_sorted_ab.txt 2>&1; tail -2 $OUT/mw_sorted_ab.txt
timeout 200 python scripts/sweep_option_ab.py update_rev 0 1 0 1 --configs=5,2 > $OUT/update_rev_ab.txt 2>&1; cat $OUT/update_rev_ab.txt
timeout 200 python scripts/execute_breakdown.py 5 > $OUT/execute_breakdown.txt 2>&1; cat $OUT/execute_breakdown.txt
timeout 400 python bench.py --gpus 8 --config 5 --steps 2 --warmup 1 --no-cpu > $OUT/bench_g8_c5.json 2>> $OUT/bench.err; cut -c1-120 $OUT/bench_g8_c5.json
{ echo "MIK_FUZZ_CASES=${FUZZ:-2000} python -m pytest tests/test_randomized_parity.py -m gpu -q -s   (MI355X, HEAD of round 4, second session: gathered row groups, points of every launch in Hilbert order, per-K-tile candidates)"; MIK_FUZZ_CASES=${FUZZ:-2000} timeout 600 python -m pytest tests/test_randomized_parity.py -m gpu -q -s 2>&1 | tail -4; } > $OUT/randomized.txt 2>&1; tail -3 $OUT/randomized.txt
rm -rf $OUT/prof5/*/*.db $OUT/ks_c5/*.db 2>/dev/null
tail -2 $OUT/bench.err
