#!/bin/bash
# End-of-round evidence, second session of round 3 (GPU-side only: the CPU legs of profiles/r03_bench_c*.json are unchanged code):
# GPU tests, bench configs 2-5 (--no-cpu, live PMC traffic), rocprofv3 kernel stats + PMC of config 2, small-problem latency,
# inverse timeline, randomized campaign.  Most important first: the call may be cut by the remaining budget.
OUT=$PWD/gpurun_out/${1:-final2}; mkdir -p $OUT; REPO=$PWD
{ nproc; free -g | head -2; rocm-smi --showproductname 2>&1 | head -8; grep -m1 "model name" /proc/cpuinfo; git -C $REPO log -1 --format=%h 2>/dev/null; } > $OUT/env.txt 2>&1
timeout 600 python -m pytest tests -m gpu -q --tb=short > $OUT/pytest_gpu.txt 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.txt; grep -E "passed|failed" $OUT/pytest_gpu.txt | tail -2
timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu > $OUT/bench_c2.json 2> $OUT/bench.err; cut -c1-200 $OUT/bench_c2.json
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks_c2 -o ks -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu --pmc off --config 2 > $OUT/ks_c2.json 2> $OUT/ks_c2.err
run() { local name=$1; shift; timeout 300 rocprofv3 "$@" --output-format csv -d $OUT/prof/$name -o $name -- python $REPO/bench.py --steps 1 --warmup 1 --no-cpu --pmc off > $OUT/prof_$name.json 2> $OUT/prof_$name.err; }
run pmc_sq --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F64
run pmc_tcc --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU TCC_HIT_sum TCC_MISS_sum
run pmc_fetch --kernel-trace --pmc FETCH_SIZE
run pmc_write --kernel-trace --pmc WRITE_SIZE
cd $REPO
python scripts/pmc_summary.py $OUT/prof > $OUT/pmc_per_kernel.csv; grep -c "k_contract" $OUT/pmc_per_kernel.csv
for c in 5 3 4; do timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu --config $c > $OUT/bench_c$c.json 2>> $OUT/bench.err; cut -c1-160 $OUT/bench_c$c.json; done
timeout 200 python scripts/small_problem_latency.py > $OUT/small_problem_latency.txt 2>&1
cd /tmp
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/tl_5000 -o tl -- python $REPO/scripts/inverse_timeline.py run 5000 > $OUT/tl_5000.txt 2>&1
cd $REPO
{ echo "== N=5000 (defaults)"; grep invert_ms $OUT/tl_5000.txt; python scripts/inverse_timeline.py parse $OUT/tl_5000; } > $OUT/inverse_timeline_final.txt 2>&1
rm -rf $OUT/tl_5000 $OUT/prof/*/*.db 2>/dev/null
{ echo "MIK_FUZZ_CASES=${FUZZ:-1000} python -m pytest tests/test_randomized_parity.py -m gpu -q -s   (MI355X, HEAD of round 3, second session: symmetrized inverse, triangular diagonal blocks, panel stream)"; MIK_FUZZ_CASES=${FUZZ:-1000} timeout 400 python -m pytest tests/test_randomized_parity.py -m gpu -q -s 2>&1 | tail -4; } > $OUT/randomized.txt 2>&1; tail -3 $OUT/randomized.txt
tail -2 $OUT/bench.err
