#!/bin/bash
# Evidence: full CPU protocol (BASELINE.md section 3 to the letter) for config 2, PMC counter summary of one bench step,
# the launcher (torch.distributed.run) form with two ranks on the one GPU, complete GPU test run.
TAG=${1:-r2i}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; REPO=$PWD
timeout 1200 python -m pytest tests -m gpu -q --tb=short -s > $OUT/pytest_gpu.txt 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.txt; tail -3 $OUT/pytest_gpu.txt; grep -n "FAILED\|Error" $OUT/pytest_gpu.txt | head
timeout 900 python bench.py --steps 5 --warmup 2 --cpu-protocol full > $OUT/bench_c2_full_cpu_protocol.json 2> $OUT/bench_full.err; python -c "
import json; d=json.load(open('$OUT/bench_c2_full_cpu_protocol.json')); cb=d['cpu_baseline']; print('BENCH', round(d['value']), cb['slab_points'], cb['protocol'], cb['kind'], cb['value'], cb['steady_state'], cb['cores'], cb.get('thread_sweep_dgemv_per_s'), cb['vectorized']['steady_state'], cb['gpu_vs_cpu_max_abs_dz'], cb['gpu_vs_cpu_max_abs_dss'])"
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --steps 2 --warmup 1 > $OUT/bench_torchrun_2ranks.json 2> $OUT/bench_torchrun_2ranks.err; echo "torchrun exit $?"; cut -c1-200 $OUT/bench_torchrun_2ranks.json; python -c "
import json; d=json.load(open('$OUT/bench_torchrun_2ranks.json')); print(d['n_gpus'], d['config']['launch'], d['config']['factor_exchange'], d['config']['factor_exchange_trial'])"
cd /tmp && export TMPDIR=/tmp
run() { local name=$1; shift; timeout 600 rocprofv3 "$@" --output-format csv -d $OUT/prof/$name -o $name -- python $REPO/bench.py --steps 1 --warmup 1 --no-cpu --pmc off > $OUT/prof_$name.json 2> $OUT/prof_$name.err; }
run pmc_sq --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F64
run pmc_lds --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU TCC_HIT_sum TCC_MISS_sum
run pmc_fetch --kernel-trace --pmc FETCH_SIZE
run pmc_write --kernel-trace --pmc WRITE_SIZE
cd $REPO; python scripts/pmc_summary.py $OUT/prof > $OUT/pmc_per_kernel.csv; grep "k_contract\|k_rhs" $OUT/pmc_per_kernel.csv | head -30
